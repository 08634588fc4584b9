"""CPU: the bare-name import surface of the reference scripts (train.py:14-16, train_speaker.py:14-16,
rational_follower.py:8) resolves to this package with ZERO edits once compat/ is on sys.path, and the
constructor signatures are the reference's."""
import inspect
import subprocess
import sys
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r'''
import sys, inspect
sys.path.insert(0, %r)
from speaker_follower_amd import compat
sys.path.insert(0, compat.path())
from env import R2RBatch, ImageFeatures                      # train.py:14
from model import EncoderLSTM, AttnDecoderLSTM               # train.py:15
from model import SpeakerEncoderLSTM, SpeakerDecoderLSTM     # train_speaker.py:15
from follower import Seq2SeqAgent                            # train.py:16
from follower import least_common_viewpoint_path, path_element_from_observation   # rational_follower.py:8
from speaker import Seq2SeqSpeaker                           # train_speaker.py:16
import model, follower, speaker, env
for m in (model, follower, speaker, env):
    assert m.__file__.startswith(compat.path()), m.__file__
def names(f): return list(inspect.signature(f).parameters)
assert names(EncoderLSTM.__init__)[1:] == ['vocab_size', 'embedding_size', 'hidden_size', 'padding_idx', 'dropout_ratio', 'bidirectional', 'num_layers', 'glove']
assert names(AttnDecoderLSTM.__init__)[1:] == ['embedding_size', 'hidden_size', 'dropout_ratio', 'feature_size', 'image_attention_layers']
assert names(SpeakerEncoderLSTM.__init__)[1:] == ['action_embedding_size', 'world_embedding_size', 'hidden_size', 'dropout_ratio', 'bidirectional']
assert names(SpeakerDecoderLSTM.__init__)[1:] == ['vocab_size', 'vocab_embedding_size', 'hidden_size', 'dropout_ratio', 'glove', 'use_input_att_feed']
assert names(Seq2SeqAgent.__init__)[1:] == ['env', 'results_path', 'encoder', 'decoder', 'episode_len', 'beam_size', 'reverse_instruction', 'max_instruction_length']
assert names(Seq2SeqSpeaker.__init__)[1:] == ['env', 'results_path', 'encoder', 'decoder', 'instruction_len', 'max_episode_len']
assert names(R2RBatch.__init__)[1:8] == ['image_features_list', 'batch_size', 'seed', 'splits', 'tokenizer', 'beam_size', 'instruction_limit']
for meth in ('train', 'test', 'rollout', 'beam_search', 'state_factored_search', '_score_obs_actions_and_instructions',
             '_rollout_with_loss', 'set_beam_size', 'save', 'load', 'write_results', '_encoder_and_decoder_paths'):
    assert callable(getattr(Seq2SeqAgent, meth)), meth
assert callable(ImageFeatures.from_args) and callable(ImageFeatures.add_args)
print('ok')
'''


def test_bare_name_imports_resolve_to_the_hip_path():
    # a fresh interpreter: the bare names `model`, `env`, ... must not leak into this test process
    res = subprocess.run([sys.executable, '-c', SCRIPT % ROOT], capture_output=True, text=True, cwd='/tmp')
    assert res.returncode == 0, res.stderr[-2000:]
    assert res.stdout.strip().endswith('ok')


def test_r2rbatch_adapter_builds_an_index_env(tmp_path):
    """R2RBatch over the committed fixture graphs: items split per instruction, tokenised, shuffled with the
    reference's seed discipline (env.py:667-699)."""
    import json
    import numpy as np
    sys.path.insert(0, ROOT)
    from speaker_follower_amd.compat import env as cenv
    conn = os.path.join(ROOT, 'tests', 'golden', 'connectivity')
    scan = sorted(f for f in os.listdir(conn) if f.endswith('_connectivity.json'))[0][:-len('_connectivity.json')]
    g = cenv.NavGraph(os.path.join(conn, scan + '_connectivity.json'))
    nodes = sorted(g.nodes())
    data = [dict(path_id=7, scan=scan, heading=0.5, path=g.path(nodes[0], nodes[-1]), distance=1.0,
                 instructions=['walk forward', 'turn left then stop', 'go'])]
    (tmp_path / 'R2R_x.json').write_text(json.dumps(data))

    class Tok:
        def encode_sentence(self, s):
            enc = np.array([4 + len(w) for w in s.split()])
            return enc, len(enc)

    class Feat:                    # stands in for MeanPooledImageFeatures: only `.store.index` is read here
        class store:
            index = {scan + '_' + n: i for i, n in enumerate(nodes)}

    envb = cenv.R2RBatch([Feat()], batch_size=2, seed=10, splits=['x'], tokenizer=Tok(), instruction_limit=2,
                         nav_graph_path=conn, data_json=str(tmp_path / 'R2R_%s.json'))
    assert sorted(it['instr_id'] for it in envb.data) == ['7_0', '7_1']
    ws = envb.reset(sort=True)
    assert len(ws) == 2 and ws[0].viewpointId == nodes[0]
    obs = envb.observe(ws)
    assert obs[0]['instr_id'] in ('7_0', '7_1') and 'adj_loc_list' in obs[0]
