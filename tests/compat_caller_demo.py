"""A caller written against the reference's PUBLIC inference API only -- the same calls, in the same order, as the
reference's demo_registration.py makes (dataset subclass with get_batch_gen / get_tf_mapping :16-110, Saver / Session /
sess.run loop :113-170, open3d RANSAC :184-192, :222-240) -- used by tests/test_compat_scripts.py on machines where the
reference checkout itself is not present (the GPU box).  Run through `python -m d3feat_amd.compat_run`; reads
demo_data/*.ply and results/Log_*/ from the working directory, writes demo_data/*.npz and registration.json."""
import json
import os

import numpy as np
import open3d
import tensorflow as tf

from datasets.common import Dataset
from models.KPFCNN_model import KernelPointFCNN
from utils.config import Config

open3d.set_verbosity_level(open3d.VerbosityLevel.Error)


class TwoClouds(Dataset):
    def __init__(self, files, voxel_size):
        Dataset.__init__(self, 'Mini')
        self.anc_points = {"train": [], "test": []}
        self.ids_list = {"train": [], "test": []}
        for f in files:
            pcd = open3d.voxel_down_sample(open3d.read_point_cloud(f), voxel_size=voxel_size)
            self.anc_points['test'].append(np.array(pcd.points))
            self.ids_list['test'].append(f)
        self.num_test = len(files)

    def get_batch_gen(self, split, config):
        def gen():
            for i in range(self.num_test):
                p = self.anc_points['test'][i].astype(np.float32)
                fid = self.ids_list['test'][i]
                yield (np.concatenate([p, p], 0), np.array([]), np.array([]), np.array([i, i], dtype=np.int32),
                       np.array([p.shape[0], p.shape[0]]), np.array([fid, fid]), np.concatenate([p, p], 0))
        return gen, (tf.float32, tf.int32, tf.int32, tf.int32, tf.int32, tf.string, tf.float32), \
            ([None, 3], [None], [None], [None], [None], [None], [None, 3])

    def get_tf_mapping(self, config):
        def tf_map(anc_points, anc_keypts, pos_keypts, obj_inds, stack_lengths, ply_id, backup_points):
            batch_inds = self.tf_get_batch_inds(stack_lengths)
            feats = tf.ones((tf.shape(anc_points)[0], 1), dtype=tf.float32)
            li = self.tf_descriptor_input(config, anc_points, feats, stack_lengths, batch_inds)
            return li + [stack_lengths, anc_keypts, pos_keypts, ply_id, backup_points]
        return tf_map


if __name__ == '__main__':
    files = ["demo_data/cloud_bin_0.ply", "demo_data/cloud_bin_1.ply"]
    path = [os.path.join('results', d) for d in sorted(os.listdir('results')) if d.startswith('Log')][-1]
    config = Config()
    config.load(path)
    dataset = TwoClouds(files, 0.03)
    dataset.init_test_input_pipeline(config)
    model = KernelPointFCNN(dataset.flat_inputs, config)
    steps = [int(f[:-5].split('-')[-1]) for f in os.listdir(os.path.join(path, 'snapshots')) if f.endswith('.meta')]
    snap = os.path.join(path, 'snapshots', 'snap-{:d}'.format(max(steps)))
    saver = tf.train.Saver(tf.get_collection(tf.GraphKeys.GLOBAL_VARIABLES, scope='KernelPointNetwork'), max_to_keep=100)
    sess = tf.Session(config=tf.ConfigProto(device_count={'GPU': 0}))
    sess.run(tf.global_variables_initializer())
    saver.restore(sess, snap)
    sess.run(dataset.test_init_op)
    for _ in range(dataset.num_test):
        inputs, features, scores, anc_id = sess.run([model.anchor_inputs, model.out_features, model.out_scores, model.anc_id],
                                                    {model.dropout_prob: 1.0})
        first = scores[inputs['in_batches'][0][:-1]]
        order = np.argsort(first, axis=0)[:].squeeze()
        np.savez_compressed(anc_id.decode("utf-8").replace('.ply', ''), keypts=inputs['backup_points'][order],
                            features=features[order], scores=scores[order])
    data = [np.load(f.replace('.ply', '.npz')) for f in files]
    pcds, feats = [], []
    for d in data:
        k = d["keypts"][-250:]
        pc = open3d.PointCloud()
        pc.points = open3d.Vector3dVector(k)
        ft = open3d.registration.Feature()
        ft.data = d["features"][-250:].T
        pcds.append(pc)
        feats.append(ft)
    result = open3d.registration_ransac_based_on_feature_matching(
        pcds[0], pcds[1], feats[0], feats[1], 0.05, open3d.TransformationEstimationPointToPoint(False), 4,
        [open3d.CorrespondenceCheckerBasedOnEdgeLength(0.9), open3d.CorrespondenceCheckerBasedOnDistance(0.05)],
        open3d.RANSACConvergenceCriteria(100000, 500))
    print(result)
    json.dump({"fitness": result.fitness, "transformation": np.asarray(result.transformation).tolist(),
               "limits": [int(x) for x in dataset.neighborhood_limits]}, open("registration.json", "w"))
