from d3feat_amd.datasets.common import Dataset, FragmentDataset, tf_batch_neighbors, tf_batch_subsampling  # noqa: F401
