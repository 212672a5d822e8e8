"""datasets.ThreeDMatch.ThreeDMatchDataset for the reference's unchanged test_3dmatch.py: the TEST split only
(datasets/ThreeDMatch.py:63-99 constructor, :326-366 prepare_geometry_registration, :138-310 generator, :312-324 mapping).
Fragments are read from data/3DMatch/fragments/<scene>/cloud_bin_*.ply and voxelised at 0.03 m; every fragment is fed
stacked with itself (anc == pos, :190-192).  Training pickles / augmentation are out of scope."""
import os
from os.path import join

import numpy as np
import open3d
import tensorflow as tf

from datasets.common import Dataset

SCENES = ['7-scenes-redkitchen', 'sun3d-home_at-home_at_scan1_2013_jan_1', 'sun3d-home_md-home_md_scan9_2012_sep_30',
          'sun3d-hotel_uc-scan3', 'sun3d-hotel_umd-maryland_hotel1', 'sun3d-hotel_umd-maryland_hotel3',
          'sun3d-mit_76_studyroom-76-1studyroom2', 'sun3d-mit_lab_hj-lab_hj_tea_nov_2_2012_scan1_erika']


class ThreeDMatchDataset(Dataset):
    def __init__(self, input_threads=8, voxel_size=0.03, load_test=False):
        Dataset.__init__(self, 'ThreeDMatch')
        self.network_model = 'descriptor'
        self.num_threads = input_threads
        self.load_test = load_test
        self.downsample = voxel_size
        self.root = 'data/3DMatch'
        self.anc_points = {'train': [], 'val': [], 'test': []}
        self.ids_list = {'train': [], 'val': [], 'test': []}
        self.num_train = self.num_val = self.num_test = 0
        if not self.load_test:
            raise NotImplementedError("ThreeDMatchDataset(load_test=False): the training split is outside the inference path")
        self.prepare_geometry_registration()

    def prepare_geometry_registration(self):
        """:326-366.  Scenes absent from data/3DMatch/fragments are skipped (the reference would stop at the first one);
        no scene at all is an error."""
        self.num_test = 0
        found = [s for s in SCENES if os.path.isdir(join(self.root, 'fragments', s))]
        if not found:
            raise FileNotFoundError("%s/fragments holds none of the 3DMatch test scenes" % self.root)
        for scene in found:
            self.test_path = join(self.root, 'fragments', scene)
            plys = sorted((f for f in os.listdir(self.test_path) if f.endswith('ply')), key=lambda x: int(x[:-4].split("_")[-1]))
            self.num_test += len(plys)
            for name in plys:
                pcd = open3d.voxel_down_sample(open3d.read_point_cloud(join(self.test_path, name)), voxel_size=0.03)
                self.anc_points['test'].append(np.array(pcd.points))
                self.ids_list['test'].append(scene + '/' + name)

    def get_batch_gen(self, split, config):
        if split != 'test':
            raise ValueError('Wrong split argument in data generator: ' + split)

        def gen():
            for i in range(self.num_test):
                pts = self.anc_points['test'][i].astype(np.float32)
                fid = self.ids_list['test'][i]
                stacked = np.concatenate([pts, pts], axis=0)
                yield (stacked, np.array([]), np.array([]), np.array([i, i], dtype=np.int32),
                       np.array([pts.shape[0], pts.shape[0]]), np.array([fid, fid]), stacked)
        gen_types = (tf.float32, tf.int32, tf.int32, tf.int32, tf.int32, tf.string, tf.float32)
        gen_shapes = ([None, 3], [None], [None], [None], [None], [None], [None, 3])
        return gen, gen_types, gen_shapes

    def get_tf_mapping(self, config):
        def tf_map(anc_points, anc_keypts, pos_keypts, obj_inds, stack_lengths, ply_id, backup_points):
            batch_inds = self.tf_get_batch_inds(stack_lengths)
            stacked_features = tf.ones((tf.shape(anc_points)[0], 1), dtype=tf.float32)
            li = self.tf_descriptor_input(config, anc_points, stacked_features, stack_lengths, batch_inds)
            return li + [stack_lengths, anc_keypts, pos_keypts, ply_id, backup_points]
        return tf_map
