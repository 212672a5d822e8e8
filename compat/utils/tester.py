"""utils.tester.ModelTester for the reference's unchanged test_3dmatch.py (utils/tester.py:136-229): same constructor, same
`generate_descriptor(model, dataset)`, same three files per fragment under
geometric_registration/D3Feat_<experiment>/{descriptors,keypoints,scores}/<scene>/ -- the session / saver objects come from
the compat `tensorflow`, every forward pass is the HIP path."""
import time

import numpy as np
import tensorflow as tf

from d3feat_amd.utils.results import save_3dmatch_results


class ModelTester:
    def __init__(self, model, restore_snap=None):
        my_vars = tf.get_collection(tf.GraphKeys.GLOBAL_VARIABLES, scope='KernelPointNetwork')
        self.saver = tf.train.Saver(my_vars, max_to_keep=100)
        cProto = tf.ConfigProto(log_device_placement=False, allow_soft_placement=True)
        cProto.gpu_options.allow_growth = True
        self.sess = tf.Session(config=cProto)
        self.sess.run(tf.global_variables_initializer())
        self.experiment_str = 'init'
        if restore_snap is not None:
            self.saver.restore(self.sess, restore_snap)
            print("Model restored from " + restore_snap)
            # '<...>_<timestamp>/snapshots/snap-<step>' -> '<timestamp[:8]>-<step>' (utils/tester.py:163)
            self.experiment_str = restore_snap.split("_")[-1][:8] + "-" + restore_snap.split("-")[-1]

    def generate_descriptor(self, model, dataset):
        """One forward pass per test fragment (self-pair); keypoints / descriptors / scores of the first cloud, rows in
        ascending score order, as .npy files (utils/tester.py:166-231)."""
        self.sess.run(dataset.test_init_op)
        self.experiment_str = self.experiment_str + '-pred'
        root = 'geometric_registration/D3Feat_%s' % self.experiment_str
        t = []
        for _ in range(dataset.num_test):
            stime = time.time()
            inputs, features, scores, anc_id = self.sess.run(
                [model.anchor_inputs, model.out_features, model.out_scores, model.anc_id], {model.dropout_prob: 1.0})
            t.append(time.time() - stime)
            first_len = int(inputs['in_batches'][0].shape[0] - 1)      # the row holds one pad entry (equal lengths)
            save_3dmatch_results(root, anc_id, inputs['backup_points'], features, scores, first_len)
            name = anc_id.decode("utf-8")
            print("Generate cloud_bin_{0} for {1}".format(int(name.split("_")[-1][:-4]), name.split("/")[0]))
            print("*" * 40)
        print("Avergae Feature Extraction Time:", np.mean(t))
