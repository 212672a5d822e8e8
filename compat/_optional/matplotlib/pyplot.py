def __getattr__(name):
    raise NotImplementedError("matplotlib is not installed; compat stand-in has no pyplot.%s" % name)
