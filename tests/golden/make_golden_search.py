#!/usr/bin/env python3
"""Golden vectors for the search procedures -- BUILD container only.

Drives the REFERENCE's Seq2SeqAgent.beam_search / state_factored_search (follower.py:541-980) and
Seq2SeqSpeaker.beam_search (speaker.py:211-318), imported from /root/reference on torch-CPU fp32,
over this repo's R2RIndexEnv (real connectivity fixtures, seeded synthetic items / features /
weights: tests/search_world.py) and stores the reference's outputs (actions, viewpoints, scores,
word ids) as tests/golden/g7_search.json.  Shims, none of which touch arithmetic:
  * stub `MatterSim` module (env.py imports the simulator at module scope),
  * `Tensor.cuda()` is the identity (follower.py:318-319 call it unconditionally),
  * uint8 masks -> bool in try_cuda (current torch rejects uint8 masks in masked_fill_),
  * env.image_features_list[0].batch_features = np.stack (env.py:330-332).
"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))

from make_golden import import_reference, load          # noqa: E402
import search_world as W                                 # noqa: E402
from speaker_follower_amd import synth                   # noqa: E402


class _Featurizer:
    def batch_features(self, feature_list):               # env.py:330-332
        return torch.from_numpy(np.stack(feature_list))


def _shim_try_cuda(x):
    return x.bool() if torch.is_tensor(x) and x.dtype == torch.uint8 else x


def cand_summary(c):
    return dict(instr_id=c['instr_id'], actions=[int(a) for a in c['actions']],
                viewpoints=[p[0] for p in c['trajectory']], score=float(c['score']),
                scores=[float(s) for s in c['scores']])


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    from speaker_follower_amd.build import build_sim
    build_sim()
    from speaker_follower_amd import sim
    real_sim = sim.load()                                 # this repo's navigation-only MatterSim
    ref_model, ref_env, ref_follower = import_reference()
    sys.modules['MatterSim'] = real_sim                   # (import_reference registers a stub)
    import speaker as ref_speaker
    torch.Tensor.cuda = lambda self, *a, **k: self
    ref_follower.try_cuda = _shim_try_cuda
    ref_speaker.try_cuda = _shim_try_cuda

    env, table = W.build_world(dense=True)
    env.image_features_list = [_Featurizer()]
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights(W.FOLLOWER_SEED)
    enc = load(ref_model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5,
                                     glove=enc_w['embedding.weight']), enc_w)
    dec = load(ref_model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat), dec_w)
    agent = ref_follower.Seq2SeqAgent(env, '', enc, dec, episode_len=W.EPISODE_LEN)
    out = dict(config=dict(scans=W.SCANS, n_items=W.N_ITEMS, batch=W.BATCH,
                           episode_len=W.EPISODE_LEN, instruction_len=W.INSTRUCTION_LEN))

    with torch.no_grad():
        out['beam'] = {}
        for beam in (1, 3, 5):
            env.set_beam_size(beam)
            env.reset_epoch()
            res = []
            for _ in range(W.N_ITEMS // W.BATCH):
                trajs, _, _ = agent.beam_search(beam)
                res += [[cand_summary(c) for c in tl] for tl in trajs]
            out['beam'][str(beam)] = res
        out['state_factored'] = {}
        for comp, succ in ((3, 1), (4, 2)):
            env.set_beam_size(max(comp, succ))
            env.reset_epoch()
            res = []
            for _ in range(W.N_ITEMS // W.BATCH):
                trajs, completed, traversed = agent.state_factored_search(comp, succ)
                res += [dict(cands=[cand_summary(c) for c in tl],
                             traversed=[s.world_state.viewpointId for s in tr])
                        for tl, tr in zip(trajs, traversed)]
            out['state_factored']['%d_%d' % (comp, succ)] = res

        # ---- speaker beam search over the gold paths of the first minibatch
        senc_w, sdec_w = synth.speaker_weights(W.SPEAKER_SEED)
        senc = load(ref_model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5), senc_w)
        sdec = load(ref_model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5,
                                                 glove=sdec_w['embedding.weight']), sdec_w)
        spk = ref_speaker.Seq2SeqSpeaker(env, '', senc, sdec, W.INSTRUCTION_LEN,
                                         max_episode_len=W.EPISODE_LEN)
        env.reset_epoch()
        path_obs, path_actions, _ = env.gold_obs_actions_and_instructions(W.EPISODE_LEN)
        out['speaker_beam'] = {}
        for beam in (1, 4):
            outs = spk.beam_search(beam, path_obs, path_actions)
            out['speaker_beam'][str(beam)] = [
                [dict(instr_id=o['instr_id'], word_indices=[int(w) for w in o['word_indices']],
                      score=float(o['score']), scores=[float(s) for s in o['scores']]) for o in ol]
                for ol in outs]

    path = os.path.join(HERE, 'g7_search.json')
    if '--big-only' not in sys.argv:
        with open(path, 'w') as f:
            json.dump(out, f)
        print('wrote', path, os.path.getsize(path), 'bytes')

    # ---- BASELINE configs[4] at its size: state-factored search, K = 40, batch 64, "peaky" weights
    # (with the default initialisation every hypothesis scores within 1e-3 of the next one and the
    # ORDER of completions is decided by roundoff)
    import time
    env, table = W.build_world(dense=True, n_items=W.BIG_ITEMS, batch=W.BIG_BATCH, item_seed=W.BIG_ITEM_SEED)
    env.image_features_list = [_Featurizer()]
    enc_w, dec_w = synth.follower_weights_peaky(W.BIG_FOLLOWER_SEED)
    enc = load(ref_model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight']), enc_w)
    dec = load(ref_model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat), dec_w)
    agent = ref_follower.Seq2SeqAgent(env, '', enc, dec, episode_len=W.BIG_EPISODE_LEN)
    env.set_beam_size(W.BIG_K)
    env.reset_epoch()
    t0 = time.time()
    with torch.no_grad():
        trajs, completed, traversed = agent.state_factored_search(W.BIG_K, 1)
    dt = time.time() - t0
    big = dict(config=dict(scans=W.SCANS, n_items=W.BIG_ITEMS, batch=W.BIG_BATCH, K=W.BIG_K,
                           episode_len=W.BIG_EPISODE_LEN, follower_seed=W.BIG_FOLLOWER_SEED,
                           reference_cpu_seconds=dt, reference_threads=torch.get_num_threads()),
               results=[dict(cands=[cand_summary(c) for c in tl],
                             traversed=[s.world_state.viewpointId for s in tr])
                        for tl, tr in zip(trajs, traversed)])
    import gzip
    path = os.path.join(HERE, 'g7_search_b64_k40.json.gz')
    with gzip.open(path, 'wt') as f:
        json.dump(big, f)
    n_c = [len(r['cands']) for r in big['results']]
    print('wrote', path, os.path.getsize(path), 'bytes; %.1f s; candidates per instance min %d mean %.1f max %d'
          % (dt, min(n_c), np.mean(n_c), max(n_c)))


if __name__ == '__main__':
    main()
