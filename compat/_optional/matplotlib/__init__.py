"""Stand-in used ONLY when matplotlib is not installed (the reference demo imports pyplot / cm without calling them)."""
