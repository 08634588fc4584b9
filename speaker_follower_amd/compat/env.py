"""`from env import R2RBatch, ImageFeatures` (train.py:14, train_speaker.py:14).

The reference's R2RBatch builds dense [36, 2176] numpy observations through 36 simulator calls per
state; the mirror is `speaker_follower_amd.env.R2RIndexEnv` (index-form observations over the HBM
feature table, same reset / observe / step / shortest-path surface).  The two adapters below give it the
reference's CONSTRUCTION surface so that train.py:make_env_and_models-style code runs as written:
`ImageFeatures.from_args(args)` -> [feature store], `R2RBatch(image_features_list, batch_size=, seed=,
splits=, tokenizer=, beam_size=, instruction_limit=)` (env.py:284-312, 667-699).  Paths are the
reference's, relative to the working directory (tasks/R2R/data/R2R_<split>.json, connectivity/,
img_features/ResNet-152-<dataset>.tsv -- or the flat .bin written by features.tsv_to_bin next to it).
"""
import json
import os

from speaker_follower_amd.env import (R2RIndexEnv, NavGraph, WorldState, make_sim, panorama_sweep,   # noqa: F401
                                      random_items)
from speaker_follower_amd.features import FeatureStore, build_loc_table, cand_sincos, tsv_to_bin     # noqa: F401

FEATURE_TSV = {'imagenet': 'img_features/ResNet-152-imagenet.tsv',
               'places365': 'img_features/ResNet-152-places365.tsv'}     # paths.py:6-9
DATA_JSON = 'tasks/R2R/data/R2R_%s.json'                                 # utils.py:54-59
CONNECTIVITY = 'connectivity'                                            # MatterSim.cpp:72


class ImageFeatures(object):
    """env.py:284-335, mean-pooled features only (the hot path's 36 x 2048 ResNet-152 table)."""
    NUM_VIEWS, MEAN_POOLED_DIM, feature_dim = 36, 2048, 2048
    IMAGE_W, IMAGE_H, VFOV = 640, 480, 60

    @staticmethod
    def add_args(argument_parser):
        argument_parser.add_argument('--image_feature_type', nargs='+', choices=['mean_pooled'], default=['mean_pooled'])
        argument_parser.add_argument('--image_attention_size', type=int)
        argument_parser.add_argument('--image_feature_datasets', nargs='+', choices=['imagenet', 'places365'],
                                     default=['imagenet'])

    @staticmethod
    def from_args(args, device='cuda'):
        kinds = sorted(getattr(args, 'image_feature_type', ['mean_pooled']))
        if kinds != ['mean_pooled']:
            raise NotImplementedError('only mean_pooled features are on the HIP path (got %s)' % kinds)
        datasets = sorted(getattr(args, 'image_feature_datasets', ['imagenet']))
        if len(datasets) != 1:
            raise NotImplementedError('one feature dataset at a time (got %s)' % datasets)
        return [MeanPooledImageFeatures(datasets, device=device)]


class MeanPooledImageFeatures(ImageFeatures):
    """env.py:350-388: holds the table as ONE HBM tensor (features.FeatureStore) instead of a dict of numpy
    arrays; `.store` is what the agents / engines take."""

    def __init__(self, image_feature_datasets, device='cuda'):
        self.image_feature_datasets = sorted(image_feature_datasets)
        tsv = FEATURE_TSV[self.image_feature_datasets[0]]
        flat = tsv[:-4] + '.bin'
        self.store = (FeatureStore.from_bin(flat, device=device) if os.path.exists(flat + '.json')
                      else FeatureStore.from_tsv(tsv, device=device))

    def get_name(self):
        return '+'.join(self.image_feature_datasets) + '_mean_pooled'

    def get_features(self, state):
        """[36, 2048] numpy block of the state's viewpoint (env.py:380-383), for dictionary-style callers."""
        row = self.store.row(state.scanId, state.location.viewpointId)
        return self.store.table[row].cpu().numpy()


class R2RBatch(R2RIndexEnv):
    """env.py:664-699 construction surface over R2RIndexEnv."""

    def __init__(self, image_features_list, batch_size=100, seed=10, splits=('train',), tokenizer=None,
                 beam_size=1, instruction_limit=None, nav_graph_path=CONNECTIVITY, data_json=DATA_JSON):
        items, self.gt = [], {}
        for split in splits:
            with open(data_json % split) as f:
                for item in json.load(f):
                    assert item['path_id'] not in self.gt
                    self.gt[item['path_id']] = item
                    instructions = item['instructions'][:instruction_limit] if instruction_limit else item['instructions']
                    for j, instr in enumerate(instructions):            # one entry per instruction
                        it = dict(item, instr_id='%s_%d' % (item['path_id'], j), instructions=instr)
                        if tokenizer:
                            it['instr_encoding'], it['instr_length'] = tokenizer.encode_sentence(instr)
                        items.append(it)
        store = image_features_list[0].store
        super().__init__(items, store.index, nav_graph_path, batch_size=batch_size, seed=seed)
        self.image_features_list = image_features_list
        self.tokenizer = tokenizer
        self.splits = list(splits)
        self.seed = seed
        self.scans = set(it['scan'] for it in items)
        self.set_beam_size(beam_size)
        self.print_progress = False
        print('R2RBatch loaded with %d instructions, using splits: %s' % (len(self.data), ','.join(splits)))
