#!/usr/bin/env python3
"""Generate the golden vectors that pin the oracle (run in the BUILD container only).

Imports the reference's tasks/R2R/{model,env,follower}.py from /root/reference on
torch-CPU fp32 (with a three-line stub `MatterSim` module, because env.py imports
the simulator at module scope, and torch.bool masks, because uint8 masks are
rejected by current torch), drives them with this repo's seeded synthetic inputs
and stores ONLY inputs' seeds / small inputs and the reference's outputs as .npz.
No reference source text is copied anywhere; the reference never travels to the
GPU box -- the .npz files do.

    python tests/golden/make_golden.py          # rewrites tests/golden/*.npz
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get('SF_REFERENCE', '/root/reference')
sys.path.insert(0, ROOT)

from speaker_follower_amd import synth            # noqa: E402
from oracle import np_env                          # noqa: E402


def import_reference():
    stub = types.ModuleType('MatterSim')
    stub.Simulator = type('Simulator', (), {})
    sys.modules['MatterSim'] = stub
    sys.path.insert(0, os.path.join(REF, 'tasks', 'R2R'))
    cwd = os.getcwd()
    os.chdir(REF)                       # env.py appends the relative 'build' dir
    try:
        import model as ref_model
        import env as ref_env
        import follower as ref_follower
    finally:
        os.chdir(cwd)
    return ref_model, ref_env, ref_follower


def load(module, state):
    module.load_state_dict({k: torch.tensor(v) for k, v in state.items()})
    module.eval()
    return module


def t(x):
    return torch.tensor(np.asarray(x))


def grads_summary(module, rng, n_samples=16):
    """Per-parameter grad L2 norm plus a few sampled entries (flat index, value)."""
    out = {}
    for name, p in module.named_parameters():
        if p.grad is None:
            continue
        g = p.grad.detach().numpy().ravel()
        idx = rng.integers(0, g.size, size=min(n_samples, g.size))
        out['gnorm/' + name] = np.float64(np.sqrt(np.sum(g.astype(np.float64) ** 2)))
        out['gidx/' + name] = idx.astype(np.int64)
        out['gval/' + name] = g[idx].astype(np.float32)
    return out


def main():
    torch.manual_seed(0)
    torch.set_num_threads(4)
    ref_model, ref_env, ref_follower = import_reference()
    out = {}

    # ---------------------------------------------------------------- G6: env helpers
    g6 = {}
    g6['loc_table'] = np.stack([ref_env.build_viewpoint_loc_embedding(v) for v in range(36)])
    rng = np.random.default_rng(66)
    feats = rng.standard_normal((36, 2048)).astype(np.float32)
    adj = [dict(absViewIndex=-1, rel_heading=0.0, rel_elevation=0.0)]
    cand_view = [0]
    cand_h = [0.0]
    cand_e = [0.0]
    for a in range(5):
        v = int(rng.integers(0, 36))
        hd = float(np.float32(rng.uniform(-np.pi, np.pi)))
        el = float(np.float32(rng.uniform(-0.5, 0.5)))
        adj.append(dict(absViewIndex=v, rel_heading=hd, rel_elevation=el))
        cand_view.append(v)
        cand_h.append(hd)
        cand_e.append(el)
    g6['act_feats'] = feats
    g6['act_cand_view'] = np.asarray(cand_view, np.int32)
    g6['act_cand_heading'] = np.asarray(cand_h, np.float32)
    g6['act_cand_elevation'] = np.asarray(cand_e, np.float32)
    g6['act_embedding'] = ref_env._build_action_embedding(adj, feats)
    instr = [list(map(int, rng.integers(4, 991, size=n))) for n in (5, 12, 85, 1, 30)]
    for tag, rev in (('fwd', False), ('rev', True)):
        seq, mask, lens = ref_follower.batch_instructions_from_encoded(instr, 80, reverse=rev)
        g6['instr_seq_' + tag] = seq.numpy()
        g6['instr_mask_' + tag] = mask.numpy().astype(bool)
        g6['instr_len_' + tag] = np.asarray(lens, np.int64)
    seq, mask, lens, perm = ref_follower.batch_instructions_from_encoded(
        instr, 80, reverse=True, sort=True)
    g6['instr_seq_sorted'] = seq.numpy()
    g6['instr_len_sorted'] = np.asarray([int(x) for x in lens], np.int64)
    g6['instr_perm_sorted'] = np.asarray([int(x) for x in perm], np.int64)
    g6['instr_tokens'] = np.asarray([x for i in instr for x in i], np.int64)
    g6['instr_sizes'] = np.asarray([len(i) for i in instr], np.int64)
    out['g6_env'] = g6

    # ---------------------------------------------------------------- G1: small-dim modules, fwd + grads
    d = synth.SMALL
    rng = np.random.default_rng(11)
    B, L, A, V = 3, 7, 4, d.views
    H, F_, D = d.hidden, d.feat, d.dot
    g1 = {}

    def rnd(*shape, scale=1.0):
        return (rng.standard_normal(shape) * scale).astype(np.float32)

    # LSTMCell
    cell = torch.nn.LSTMCell(2 * F_, H)
    x, h, c = rnd(B, 2 * F_), rnd(B, H), rnd(B, H)
    xt, ht, ct = (t(a).requires_grad_(True) for a in (x, h, c))
    h1, c1 = cell(xt, (ht, ct))
    gh, gc = rnd(B, H), rnd(B, H)
    (h1 * t(gh)).sum().add((c1 * t(gc)).sum()).backward()
    g1.update({'lstm/' + k: v.detach().numpy() for k, v in cell.state_dict().items()})
    g1.update({'lstm/x': x, 'lstm/h': h, 'lstm/c': c, 'lstm/gh': gh, 'lstm/gc': gc,
               'lstm/h1': h1.detach().numpy(), 'lstm/c1': c1.detach().numpy(),
               'lstm/dx': xt.grad.numpy(), 'lstm/dh': ht.grad.numpy(), 'lstm/dc': ct.grad.numpy()})
    g1.update({'lstm/d_' + k: p.grad.numpy() for k, p in cell.named_parameters()})

    # SoftDotAttention
    att = ref_model.SoftDotAttention(H)
    hh, ctx = rnd(B, H), rnd(B, L, H)
    mask = np.zeros((B, L), bool)
    mask[0, 5:] = True
    mask[2, 3:] = True
    ht, ctxt = t(hh).requires_grad_(True), t(ctx).requires_grad_(True)
    h_tilde, alpha = att(ht, ctxt, t(mask))
    go = rnd(B, H)
    (h_tilde * t(go)).sum().backward()
    g1.update({'sda/' + k: v.detach().numpy() for k, v in att.state_dict().items()})
    g1.update({'sda/h': hh, 'sda/ctx': ctx, 'sda/mask': mask, 'sda/go': go,
               'sda/h_tilde': h_tilde.detach().numpy(), 'sda/alpha': alpha.detach().numpy(),
               'sda/dh': ht.grad.numpy(), 'sda/dctx': ctxt.grad.numpy()})
    g1.update({'sda/d_' + k: p.grad.numpy() for k, p in att.named_parameters()})

    # VisualSoftDotAttention
    vat = ref_model.VisualSoftDotAttention(H, F_, dot_dim=D)
    hh, X = rnd(B, H), rnd(B, V, F_)
    ht, Xt = t(hh).requires_grad_(True), t(X).requires_grad_(True)
    wctx, al = vat(ht, Xt)
    go = rnd(B, F_)
    (wctx * t(go)).sum().backward()
    g1.update({'vsda/' + k: v.detach().numpy() for k, v in vat.state_dict().items()})
    g1.update({'vsda/h': hh, 'vsda/X': X, 'vsda/go': go,
               'vsda/out': wctx.detach().numpy(), 'vsda/alpha': al.detach().numpy(),
               'vsda/dh': ht.grad.numpy(), 'vsda/dX': Xt.grad.numpy()})
    g1.update({'vsda/d_' + k: p.grad.numpy() for k, p in vat.named_parameters()})

    # EltwiseProdScoring
    eps = ref_model.EltwiseProdScoring(H, F_, dot_dim=D)
    hh, U = rnd(B, H), rnd(B, A, F_)
    ht, Ut = t(hh).requires_grad_(True), t(U).requires_grad_(True)
    lg = eps(ht, Ut)
    go = rnd(B, A)
    (lg * t(go)).sum().backward()
    g1.update({'eps/' + k: v.detach().numpy() for k, v in eps.state_dict().items()})
    g1.update({'eps/h': hh, 'eps/U': U, 'eps/go': go, 'eps/logit': lg.detach().numpy(),
               'eps/dh': ht.grad.numpy(), 'eps/dU': Ut.grad.numpy()})
    g1.update({'eps/d_' + k: p.grad.numpy() for k, p in eps.named_parameters()})
    out['g1_modules_small'] = g1

    # ---------------------------------------------------------------- follower, full dims
    dims = synth.FULL
    enc_w, dec_w = synth.follower_weights(101, dims)
    enc = ref_model.EncoderLSTM(dims.vocab, dims.word, dims.hidden, 0, 0.5,
                                glove=enc_w['embedding.weight'])
    dec = ref_model.AttnDecoderLSTM(dims.feat, dims.hidden, 0.5, feature_size=dims.feat)
    load(enc, enc_w)
    load(dec, dec_w)
    loc_table = np_env.static_loc_embeddings()

    def encode(fb, max_len=80):
        seq, mask, lens = np_env.batch_instructions_from_encoded(fb.instr, max_len, reverse=True)
        return seq, mask, lens

    def ref_rollout(fb, table, steps, feedback, with_grad):
        """follower.py:430-539 driven over precomputed observations with the
        REFERENCE modules (the agent itself cannot run: hard .cuda(), live simulator)."""
        seq, mask, lens = encode(fb)
        ctx, h, c = enc(t(seq), lens)
        B = seq.shape[0]
        u_prev = dec.u_begin.expand(B, -1)
        ended = np.zeros(B, bool)
        crit = torch.nn.CrossEntropyLoss(ignore_index=-1)
        loss = 0
        scores = torch.zeros(B)
        logits, actions, alphas_v, alphas = [], [], [], []
        for st in range(steps):
            X, all_u, is_valid = np_env.dense_follower_step(table, loc_table, fb, st)
            all_u_t = t(all_u)
            h, c, alpha, logit, alpha_v = dec(u_prev, all_u_t, t(X), h, c, ctx, t(mask))
            logit[t(is_valid) == 0] = -float('inf')
            target = t(np.where(ended, -1, fb.target[st]))
            if (target != -1).any():
                loss = loss + crit(logit, target)
            if feedback == 'teacher':
                a_t = torch.clamp(target, min=0)
            else:
                _, a_t = logit.max(1)
                a_t = a_t.detach()
            u_prev = all_u_t[np.arange(B), a_t, :].detach()
            scores += -torch.nn.functional.cross_entropy(logit, a_t, reduction='none').data
            logits.append(logit.detach().numpy().copy())
            actions.append(a_t.numpy().copy())
            alphas_v.append(alpha_v.detach().numpy().copy())
            alphas.append(alpha.detach().numpy().copy())
            ended |= (a_t.numpy() == 0)
            if ended.all():
                break
        res = dict(actions=np.stack(actions), loss=np.float32(float(loss)),
                   scores=scores.numpy(), h=h.detach().numpy(), c=c.detach().numpy(),
                   n_steps=np.int64(len(logits)))
        A = fb.a_max
        lg = np.full((len(logits), B, A), -np.inf, np.float32)
        for i, l in enumerate(logits):
            lg[i, :, :l.shape[1]] = l
        res['logits'] = lg
        res['alpha_v'] = np.stack(alphas_v)
        res['alpha_last'] = alphas[-1]
        if with_grad:
            enc.zero_grad()
            dec.zero_grad()
            loss.backward()
            grng = np.random.default_rng(404)
            res.update({'enc/' + k: v for k, v in grads_summary(enc, grng).items()})
            res.update({'dec/' + k: v for k, v in grads_summary(dec, grng).items()})
        return res

    # G2 + G3: single decoder step and encoder at config-1 shapes (B=8)
    fb8 = synth.follower_batch(seed=7, batch=8, steps=10, n_viewpoints=64, min_len=3,
                               max_len=19, a_max=8)
    table64 = synth.feature_table(7, 64)
    seq, mask, lens = encode(fb8, 80)
    with torch.no_grad():
        ctx, h0, c0 = enc(t(seq), lens)
        X, all_u, is_valid = np_env.dense_follower_step(table64, loc_table, fb8, 0)
        rng = np.random.default_rng(22)
        u_prev = all_u[np.arange(8), rng.integers(0, fb8.a_num[0])]
        h1, c1, alpha, logit, alpha_v = dec(t(u_prev), t(all_u), t(X), h0, c0, ctx, t(mask))
    out['g3_encoder'] = dict(seed=np.int64(7), ctx=ctx.numpy(), decoder_init=h0.numpy(),
                             c_t=c0.numpy(), lengths=np.asarray(lens, np.int64))
    out['g2_decoder_step'] = dict(seed=np.int64(7), u_prev=u_prev, h1=h1.numpy(), c1=c1.numpy(),
                                  alpha=alpha.numpy(), logit=logit.numpy(),
                                  alpha_v=alpha_v.numpy())

    # G4: rollouts.  B=8: 10-step teacher (with grads) and argmax; B=100: 20-step argmax + teacher grads
    out['g4_rollout_b8_teacher'] = ref_rollout(fb8, table64, 10, 'teacher', True)
    with torch.no_grad():
        out['g4_rollout_b8_argmax'] = ref_rollout(fb8, table64, 10, 'argmax', False)
    fb100 = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=256)
    table256 = synth.feature_table(0, 256)
    with torch.no_grad():
        out['g4_rollout_b100_argmax'] = ref_rollout(fb100, table256, 20, 'argmax', False)
    out['g4_rollout_b100_teacher'] = ref_rollout(fb100, table256, 20, 'teacher', True)

    # ---------------------------------------------------------------- G5: speaker
    senc_w, sdec_w = synth.speaker_weights(202, dims)
    senc = ref_model.SpeakerEncoderLSTM(dims.feat, dims.feat, dims.hidden, 0.5)
    sdec = ref_model.SpeakerDecoderLSTM(dims.vocab, dims.word, dims.hidden, 0.5,
                                        glove=sdec_w['embedding.weight'])
    load(senc, senc_w)
    load(sdec, sdec_w)

    def ref_speaker(sb, table, steps, feedback, with_grad):
        acts, feats, path_mask = np_env.dense_speaker_inputs(sb, table, loc_table)
        instr_seq, _, _ = np_env.batch_instructions_from_encoded(sb.instr, 80)
        ctx, h, c = senc([t(a) for a in acts], [t(f) for f in feats])
        B = ctx.shape[0]
        w_t = torch.full((B,), 3, dtype=torch.long)
        ended = np.zeros(B, bool)
        loss = 0
        scores = torch.zeros(B)
        words, logits = [], []
        for st in range(steps):
            h, c, alpha, logit = sdec(w_t.view(-1, 1), h, c, ctx, t(path_mask))
            target = t(instr_seq[:, st]).contiguous()
            if feedback == 'teacher':
                w_t = target
            else:
                _, w_t = logit.max(1)
                w_t = w_t.detach()
            logp = torch.nn.functional.log_softmax(logit, dim=1)
            scores += -torch.nn.functional.nll_loss(logp, w_t, ignore_index=0,
                                                    reduction='none').data
            if (target != 0).any():
                loss = loss + torch.nn.functional.nll_loss(logp, target, ignore_index=0)
            logits.append(logit.detach().numpy().copy())
            words.append(w_t.numpy().copy())
            ended |= (w_t.numpy() == 2)
            if ended.all():
                break
        res = dict(words=np.stack(words), loss=np.float32(float(loss)), scores=scores.numpy(),
                   ctx=ctx.detach().numpy(), h=h.detach().numpy(), c=c.detach().numpy(),
                   logits_first=np.stack(logits[:3]), logit_last=logits[-1],
                   n_steps=np.int64(len(logits)))
        if with_grad:
            senc.zero_grad()
            sdec.zero_grad()
            loss.backward()
            grng = np.random.default_rng(505)
            res.update({'enc/' + k: v for k, v in grads_summary(senc, grng).items()})
            res.update({'dec/' + k: v for k, v in grads_summary(sdec, grng).items()})
        return res

    sb6 = synth.speaker_batch(seed=9, batch=6, n_viewpoints=64, min_len=3, max_len=25)
    out['g5_speaker_b6_teacher'] = ref_speaker(sb6, table64, 80, 'teacher', True)
    with torch.no_grad():
        out['g5_speaker_b6_argmax'] = ref_speaker(sb6, table64, 30, 'argmax', False)

    for name, arrays in out.items():
        path = os.path.join(HERE, name + '.npz')
        with tempfile.NamedTemporaryFile(dir=HERE, suffix='.npz', delete=False) as f:
            np.savez_compressed(f, **arrays)
        os.replace(f.name, path)
        print('%-28s %8.1f KB  %d arrays' % (name, os.path.getsize(path) / 1024, len(arrays)))


if __name__ == '__main__':
    main()
