"""Seeded synthetic R2R-shaped inputs and weights (numpy PCG64, host side only).

The shapes follow SURVEY.md section 8(d): S1 (plumbing), S2 (headline follower
rollout, batch 100, 36 views x 2048-d, <=80-token instructions, 20 decode
steps) and S3 (speaker).  Everything here is plain numpy so that the very same
arrays feed the oracle (tests / cpu_baseline) and, after upload, the HIP path.

Reference shapes this mirrors (read-only citations into the reference tree):
  * feature store 36 x 2048 fp32 per viewpoint       tasks/R2R/env.py:350-383
  * candidate list = stop + neighbours                tasks/R2R/env.py:60-75, 149-224
  * instruction encoding (no BOS/EOS, PAD=0, EOS=2)   tasks/R2R/follower.py:75-105
  * model sizes                                       tasks/R2R/train.py:26-40
"""
from collections import OrderedDict
from dataclasses import dataclass

import numpy as np

PAD, UNK, EOS, BOS = 0, 1, 2, 3     # tasks/R2R/utils.py:19-24


@dataclass(frozen=True)
class Dims:
    hidden: int = 512           # train.py:33
    img: int = 2048             # env.py:286
    loc: int = 128              # env.py:62, 87
    dot: int = 256              # model.py:303, 335
    word: int = 300             # train.py:30
    vocab: int = 991            # tasks/R2R/data/train_vocab.txt (+4 base tokens)
    views: int = 36             # env.py:285

    @property
    def feat(self):
        return self.img + self.loc


FULL = Dims()
# Small dims used by the per-module golden vectors.  `loc` stays a
# multiple of 4 so that the four sin/cos groups exist.
SMALL = Dims(hidden=16, img=16, loc=8, dot=8, word=12, vocab=23, views=5)


def _uniform(rng, shape, k):
    return rng.uniform(-k, k, size=shape).astype(np.float32)


def follower_weights(seed, dims=FULL):
    """(encoder_state, decoder_state): OrderedDicts keyed like the reference
    state_dicts (model.py:55-65 and :371-375)."""
    rng = np.random.default_rng(seed)
    H, E, F, D, V = dims.hidden, dims.word, dims.feat, dims.dot, dims.vocab
    kh = 1.0 / np.sqrt(H)
    enc = OrderedDict()
    emb = (rng.standard_normal((V, E)) * 0.4).astype(np.float32)
    emb[PAD] = 0.0                                   # nn.Embedding padding_idx row
    enc['embedding.weight'] = emb
    enc['lstm.weight_ih_l0'] = _uniform(rng, (4 * H, E), kh)
    enc['lstm.weight_hh_l0'] = _uniform(rng, (4 * H, H), kh)
    enc['lstm.bias_ih_l0'] = _uniform(rng, (4 * H,), kh)
    enc['lstm.bias_hh_l0'] = _uniform(rng, (4 * H,), kh)
    enc['encoder2decoder.weight'] = _uniform(rng, (H, H), kh)
    enc['encoder2decoder.bias'] = _uniform(rng, (H,), kh)

    dec = OrderedDict()
    dec['lstm.weight_ih'] = _uniform(rng, (4 * H, 2 * F), kh)
    dec['lstm.weight_hh'] = _uniform(rng, (4 * H, H), kh)
    dec['lstm.bias_ih'] = _uniform(rng, (4 * H,), kh)
    dec['lstm.bias_hh'] = _uniform(rng, (4 * H,), kh)
    dec['visual_attention_layer.linear_in_h.weight'] = _uniform(rng, (D, H), kh)
    dec['visual_attention_layer.linear_in_h.bias'] = _uniform(rng, (D,), kh)
    kf = 1.0 / np.sqrt(F)
    dec['visual_attention_layer.linear_in_v.weight'] = _uniform(rng, (D, F), kf)
    dec['visual_attention_layer.linear_in_v.bias'] = _uniform(rng, (D,), kf)
    dec['text_attention_layer.linear_in.weight'] = _uniform(rng, (H, H), kh)
    dec['text_attention_layer.linear_out.weight'] = _uniform(rng, (H, 2 * H), 1.0 / np.sqrt(2 * H))
    dec['decoder2action.linear_in_h.weight'] = _uniform(rng, (D, H), kh)
    dec['decoder2action.linear_in_h.bias'] = _uniform(rng, (D,), kh)
    dec['decoder2action.linear_in_a.weight'] = _uniform(rng, (D, F), kf)
    dec['decoder2action.linear_in_a.bias'] = _uniform(rng, (D,), kf)
    kd = 1.0 / np.sqrt(D)
    dec['decoder2action.linear_out.weight'] = _uniform(rng, (1, D), kd)
    dec['decoder2action.linear_out.bias'] = _uniform(rng, (1,), kd)
    return enc, dec


# Gains that turn the torch-default initialisation above (logit spread ~5e-3, near-uniform
# attention) into a regime like a trained model's: visual attention max ~0.8, text attention max
# ~0.8, action logits with O(1) spread.  Parity on these weights has teeth: an error in an
# attention or scoring kernel moves logits by far more than the tolerance.
PEAKY_GAINS = {
    'enc': {'lstm.weight_ih_l0': 3.0, 'lstm.weight_hh_l0': 2.0},
    'dec': {'visual_attention_layer.linear_in_h.weight': 4.0,
            'visual_attention_layer.linear_in_v.weight': 4.0,
            'text_attention_layer.linear_in.weight': 40.0,
            'decoder2action.linear_in_h.weight': 6.0,
            'decoder2action.linear_in_a.weight': 6.0,
            'decoder2action.linear_out.weight': 8.0},
}


def follower_weights_peaky(seed, dims=FULL):
    """`follower_weights(seed)` with PEAKY_GAINS applied (same keys, same shapes)."""
    enc, dec = follower_weights(seed, dims)
    for k, g in PEAKY_GAINS['enc'].items():
        enc[k] = (enc[k] * np.float32(g)).astype(np.float32)
    for k, g in PEAKY_GAINS['dec'].items():
        dec[k] = (dec[k] * np.float32(g)).astype(np.float32)
    return enc, dec


def bidirectional_encoder_weights(seed, dims=FULL):
    """State of EncoderLSTM(..., hidden_size // 2, bidirectional=True) (model.py:47-66; train.py:197-199): the two
    directions' LSTM weights at hidden // 2 each (with the PEAKY_GAINS of the unidirectional encoder) and
    encoder2decoder over the concatenated state."""
    rng = np.random.default_rng(seed)
    Hd, E, V = dims.hidden // 2, dims.word, dims.vocab
    k = 1.0 / np.sqrt(Hd)
    enc = OrderedDict()
    emb = (rng.standard_normal((V, E)) * 0.4).astype(np.float32)
    emb[PAD] = 0.0
    enc['embedding.weight'] = emb
    for sfx in ('', '_reverse'):
        enc['lstm.weight_ih_l0' + sfx] = _uniform(rng, (4 * Hd, E), k) * np.float32(3.0)
        enc['lstm.weight_hh_l0' + sfx] = _uniform(rng, (4 * Hd, Hd), k) * np.float32(2.0)
        enc['lstm.bias_ih_l0' + sfx] = _uniform(rng, (4 * Hd,), k)
        enc['lstm.bias_hh_l0' + sfx] = _uniform(rng, (4 * Hd,), k)
    k2 = 1.0 / np.sqrt(2 * Hd)
    enc['encoder2decoder.weight'] = _uniform(rng, (2 * Hd, 2 * Hd), k2)
    enc['encoder2decoder.bias'] = _uniform(rng, (2 * Hd,), k2)
    return enc


def speaker_weights_peaky(seed, dims=FULL):
    """`speaker_weights(seed)` with attention and output gains (path attention max ~0.7, word
    distribution with a clear mode)."""
    enc, dec = speaker_weights(seed, dims)
    for k, g in (('visual_attention_layer.linear_in_h.weight', 8.0),
                 ('visual_attention_layer.linear_in_v.weight', 8.0)):
        enc[k] = (enc[k] * np.float32(g)).astype(np.float32)
    for k, g in (('attention_layer.linear_in.weight', 25.0), ('decoder2action.weight', 60.0),
                 ('lstm.weight_ih', 3.0), ('lstm.weight_hh', 2.0)):
        dec[k] = (dec[k] * np.float32(g)).astype(np.float32)
    return enc, dec


def speaker_weights(seed, dims=FULL):
    """(encoder_state, decoder_state) keyed like SpeakerEncoderLSTM / SpeakerDecoderLSTM
    (model.py:415-419 and :467-485)."""
    rng = np.random.default_rng(seed)
    H, E, F, D, V = dims.hidden, dims.word, dims.feat, dims.dot, dims.vocab
    kh = 1.0 / np.sqrt(H)
    kf = 1.0 / np.sqrt(F)
    enc = OrderedDict()
    enc['visual_attention_layer.linear_in_h.weight'] = _uniform(rng, (D, H), kh)
    enc['visual_attention_layer.linear_in_h.bias'] = _uniform(rng, (D,), kh)
    enc['visual_attention_layer.linear_in_v.weight'] = _uniform(rng, (D, F), kf)
    enc['visual_attention_layer.linear_in_v.bias'] = _uniform(rng, (D,), kf)
    enc['lstm.weight_ih'] = _uniform(rng, (4 * H, 2 * F), kh)
    enc['lstm.weight_hh'] = _uniform(rng, (4 * H, H), kh)
    enc['lstm.bias_ih'] = _uniform(rng, (4 * H,), kh)
    enc['lstm.bias_hh'] = _uniform(rng, (4 * H,), kh)
    enc['encoder2decoder.weight'] = _uniform(rng, (H, H), kh)
    enc['encoder2decoder.bias'] = _uniform(rng, (H,), kh)
    dec = OrderedDict()
    dec['embedding.weight'] = (rng.standard_normal((V, E)) * 0.4).astype(np.float32)
    dec['lstm.weight_ih'] = _uniform(rng, (4 * H, E), kh)
    dec['lstm.weight_hh'] = _uniform(rng, (4 * H, H), kh)
    dec['lstm.bias_ih'] = _uniform(rng, (4 * H,), kh)
    dec['lstm.bias_hh'] = _uniform(rng, (4 * H,), kh)
    dec['attention_layer.linear_in.weight'] = _uniform(rng, (H, H), kh)
    dec['attention_layer.linear_out.weight'] = _uniform(rng, (H, 2 * H), 1.0 / np.sqrt(2 * H))
    dec['decoder2action.weight'] = _uniform(rng, (V, H), kh)
    dec['decoder2action.bias'] = _uniform(rng, (V,), kh)
    return enc, dec


def speaker_decoder_att_feed_weights(seed, dims=FULL):
    """State of SpeakerDecoderLSTM(use_input_att_feed=True) (model.py:475-485): LSTMCell(E + H -> H),
    ContextOnlySoftDotAttention.linear_in, output_l1 (2H -> H), decoder2action; gains as in `speaker_weights_peaky`."""
    rng = np.random.default_rng([seed, 0xA77F])
    H, E, V = dims.hidden, dims.word, dims.vocab
    kh = 1.0 / np.sqrt(H)
    dec = OrderedDict()
    dec['embedding.weight'] = (rng.standard_normal((V, E)) * 0.4).astype(np.float32)
    dec['lstm.weight_ih'] = (_uniform(rng, (4 * H, E + H), kh) * np.float32(2.0)).astype(np.float32)
    dec['lstm.weight_hh'] = (_uniform(rng, (4 * H, H), kh) * np.float32(2.0)).astype(np.float32)
    dec['lstm.bias_ih'] = _uniform(rng, (4 * H,), kh)
    dec['lstm.bias_hh'] = _uniform(rng, (4 * H,), kh)
    dec['attention_layer.linear_in.weight'] = (_uniform(rng, (H, H), kh) * np.float32(20.0)).astype(np.float32)
    dec['output_l1.weight'] = (_uniform(rng, (H, 2 * H), 1.0 / np.sqrt(2 * H)) * np.float32(3.0)).astype(np.float32)
    dec['output_l1.bias'] = _uniform(rng, (H,), 1.0 / np.sqrt(2 * H))
    dec['decoder2action.weight'] = (_uniform(rng, (V, H), kh) * np.float32(20.0)).astype(np.float32)
    dec['decoder2action.bias'] = _uniform(rng, (V,), kh)
    return dec


def feature_table(seed, n_viewpoints, dims=FULL):
    """[n_viewpoints, views, img] fp32, ResNet-pool5-like: 0.5*N(0,1) clipped at 0."""
    rng = np.random.default_rng([seed, 0x7AB1E])
    t = rng.standard_normal((n_viewpoints, dims.views, dims.img), dtype=np.float32)
    t *= 0.5
    np.maximum(t, 0.0, out=t)
    return t


def instructions(seed, batch, min_len, max_len, dims=FULL, sort=True):
    """Ragged token lists (no BOS/EOS), lengths sorted descending like
    R2RBatch._next_minibatch(sort=True) (env.py:733-734)."""
    rng = np.random.default_rng([seed, 0x1257])
    lens = rng.integers(min_len, max_len + 1, size=batch)
    if sort:
        lens = np.sort(lens)[::-1]
    return [rng.integers(4, dims.vocab, size=int(n)).astype(np.int64) for n in lens]


@dataclass
class FollowerBatch:
    """Index form of one follower episode batch (what the HIP path consumes).

    Per decode step t and sample b the agent stands at viewpoint row `vp[t,b]`
    of the feature table, facing discretised view `view[t,b]`; candidate a has
    feature row (cand_vp, cand_view) plus sin/cos of its relative angles.
    Candidate 0 is the stop action (all-zero embedding, env.py:64-66)."""
    instr: list            # ragged int64 token lists, len B
    vp: np.ndarray         # [T,B] int32 row into the feature table
    view: np.ndarray       # [T,B] int32 in [0,views)
    a_num: np.ndarray      # [T,B] int32, 2..a_max (incl. stop)
    cand_view: np.ndarray  # [T,B,A] int32 absolute view index of candidate (row of vp's panorama)
    cand_heading: np.ndarray    # [T,B,A] float32 rel_heading
    cand_elevation: np.ndarray  # [T,B,A] float32 rel_elevation
    target: np.ndarray     # [T,B] int64 teacher action, -1 once ended
    a_max: int


def follower_batch(seed, batch, steps, n_viewpoints, min_len=10, max_len=79,
                   a_max=14, dims=FULL, stop_prob=1.0 / 6.0):
    """S2-style batch: next viewpoint drawn uniformly (no simulator),
    a_num ~ 1 + clip(Poisson(4), 1, a_max-1), targets uniform over valid
    candidates with a geometric stop time, -1 afterwards (follower.py:327)."""
    rng = np.random.default_rng([seed, 0xF0110])
    T, B, A = steps, batch, a_max
    vp = rng.integers(0, n_viewpoints, size=(T, B)).astype(np.int32)
    view = rng.integers(0, dims.views, size=(T, B)).astype(np.int32)
    a_num = (1 + np.clip(rng.poisson(4.0, size=(T, B)), 1, A - 1)).astype(np.int32)
    cand_view = rng.integers(0, dims.views, size=(T, B, A)).astype(np.int32)
    cand_heading = rng.uniform(-np.pi, np.pi, size=(T, B, A)).astype(np.float32)
    cand_elevation = rng.uniform(-np.pi / 6, np.pi / 6, size=(T, B, A)).astype(np.float32)
    target = np.empty((T, B), np.int64)
    ended = np.zeros(B, bool)
    for t in range(T):
        stop = rng.random(B) < stop_prob
        move = 1 + (rng.random(B) * (a_num[t] - 1)).astype(np.int64)
        move = np.minimum(move, a_num[t] - 1)
        tgt = np.where(stop, 0, move)
        target[t] = np.where(ended, -1, tgt)
        ended |= (tgt == 0)
    # at least the first step is always live (loop would otherwise break, follower.py:533)
    instr = instructions(seed, B, min_len, max_len, dims)
    return FollowerBatch(instr, vp, view, a_num, cand_view, cand_heading,
                         cand_elevation, target, A)


@dataclass
class SpeakerBatch:
    """Index form of one speaker batch: per path step the panorama the agent saw
    and the action it took (speaker.py:68-121), plus the target instruction."""
    instr: list            # ragged int64 token lists (targets, not reversed: speaker.py:131)
    path_len: np.ndarray   # [B] int32 number of actions (incl. final stop)
    vp: np.ndarray         # [Tp,B]
    view: np.ndarray       # [Tp,B]
    act_view: np.ndarray   # [Tp,B] absolute view index of the chosen candidate
    act_heading: np.ndarray     # [Tp,B]
    act_elevation: np.ndarray   # [Tp,B]
    act_is_stop: np.ndarray     # [Tp,B] bool: chosen action was stop (zero embedding)


def speaker_batch(seed, batch, n_viewpoints, min_path=4, max_path=7, min_len=10,
                  max_len=79, dims=FULL):
    rng = np.random.default_rng([seed, 0x5BEA])
    B = batch
    path_len = rng.integers(min_path, max_path + 1, size=B).astype(np.int32)
    Tp = int(path_len.max())
    vp = rng.integers(0, n_viewpoints, size=(Tp, B)).astype(np.int32)
    view = rng.integers(0, dims.views, size=(Tp, B)).astype(np.int32)
    act_view = rng.integers(0, dims.views, size=(Tp, B)).astype(np.int32)
    act_heading = rng.uniform(-np.pi, np.pi, size=(Tp, B)).astype(np.float32)
    act_elevation = rng.uniform(-np.pi / 6, np.pi / 6, size=(Tp, B)).astype(np.float32)
    steps = np.arange(Tp)[:, None]
    act_is_stop = steps == (path_len[None, :] - 1)      # last action of each path is stop
    instr = instructions(seed, B, min_len, max_len, dims, sort=False)
    return SpeakerBatch(instr, path_len, vp, view, act_view, act_heading,
                        act_elevation, act_is_stop)
