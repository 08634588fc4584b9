"""Oracle (test infrastructure): numpy restatement of the reference's host-side
feature assembly and instruction batching.  See oracle/__init__.py for the rules.

Each function cites the reference lines it follows (paths under /root/reference).
"""
import numpy as np

PAD, UNK, EOS, BOS = 0, 1, 2, 3          # tasks/R2R/utils.py:19-24
ANGLE_INC = np.pi / 6.0                   # tasks/R2R/env.py:57


def viewpoint_loc_embedding(view_index, n_views=36, loc=128):
    """tasks/R2R/env.py:78-96 build_viewpoint_loc_embedding.

    Row v (absolute view index) holds sin/cos of the heading and elevation of
    view v *relative to the agent's current view*; 12 headings x 3 elevations."""
    g = loc // 4
    emb = np.zeros((n_views, loc), np.float32)
    for abs_view in range(n_views):
        rel = (abs_view - view_index) % 12 + (abs_view // 12) * 12
        rel_heading = (rel % 12) * ANGLE_INC
        rel_elevation = (rel // 12 - 1) * ANGLE_INC
        emb[abs_view, 0:g] = np.sin(rel_heading)
        emb[abs_view, g:2 * g] = np.cos(rel_heading)
        emb[abs_view, 2 * g:3 * g] = np.sin(rel_elevation)
        emb[abs_view, 3 * g:] = np.cos(rel_elevation)
    return emb


def static_loc_embeddings(n_views=36, loc=128):
    """tasks/R2R/env.py:100-101: one table per possible agent view index."""
    return np.stack([viewpoint_loc_embedding(v, n_views, loc) for v in range(n_views)])


def panorama_feature(features, view_index, loc_table):
    """tasks/R2R/env.py:773: concat(features[36,img], loc_embedding[viewIndex])."""
    return np.concatenate((features, loc_table[view_index]), axis=-1)


def action_embedding(features, cand_view, cand_heading, cand_elevation, loc=128):
    """tasks/R2R/env.py:60-75 _build_action_embedding.

    features [views,img]; candidate 0 is 'stop' and stays zero; candidate a>0 is
    features[absViewIndex] || sin/cos(rel_heading) || sin/cos(rel_elevation)."""
    n = len(cand_view)
    img = features.shape[-1]
    g = loc // 4
    emb = np.zeros((n, img + loc), np.float32)
    for a in range(1, n):
        emb[a, :img] = features[cand_view[a]]
        le = emb[a, img:]
        # the simulator hands over python floats: sin/cos in float64, then stored as fp32
        hd, el = np.float64(cand_heading[a]), np.float64(cand_elevation[a])
        le[0:g] = np.sin(hd)
        le[g:2 * g] = np.cos(hd)
        le[2 * g:3 * g] = np.sin(el)
        le[3 * g:] = np.cos(el)
    return emb


def action_variable(action_embeddings):
    """tasks/R2R/follower.py:300-320 _action_variable: zero-pad to the batch's
    max candidate count; is_valid marks real candidates."""
    max_a = max(len(e) for e in action_embeddings)
    dim = action_embeddings[0].shape[-1]
    B = len(action_embeddings)
    is_valid = np.zeros((B, max_a), np.float32)
    all_u = np.zeros((B, max_a, dim), np.float32)
    for i, e in enumerate(action_embeddings):
        is_valid[i, :len(e)] = 1.0
        all_u[i, :len(e)] = e
    return all_u, is_valid


def batch_instructions_from_encoded(encoded, max_length, reverse=False, sort=False):
    """tasks/R2R/follower.py:75-105.  Returns (seq[B,max_length] int64,
    mask[B,max(len)] bool (True = PAD), lengths list[, perm])."""
    n = len(encoded)
    seq = np.full((n, max_length), PAD, np.int64)
    lengths = []
    for i, inst in enumerate(encoded):
        inst = np.asarray(inst, np.int64)
        if len(inst) > 0:
            assert inst[-1] != EOS
        if reverse:
            inst = inst[::-1]
        inst = np.concatenate((inst, [EOS]))[:max_length]
        seq[i, :len(inst)] = inst
        lengths.append(len(inst))
    perm = None
    if sort:
        perm = np.argsort(-np.asarray(lengths), kind='stable')
        lengths = [lengths[i] for i in perm]
        seq = seq[perm]
    mask = (seq == PAD)[:, :max(lengths)]
    if sort:
        return seq, mask, lengths, list(perm)
    return seq, mask, lengths


def dense_follower_step(table, loc_table, fb, t, loc=128):
    """Dense tensors the reference agent would build for decode step t of a
    synth.FollowerBatch: X [B,views,F] (follower.py:291-298 + env.py:771-773),
    all_u [B,A,F], is_valid [B,A] (follower.py:300-320)."""
    B = fb.vp.shape[1]
    X = np.stack([panorama_feature(table[fb.vp[t, b]], fb.view[t, b], loc_table)
                  for b in range(B)])
    embs = []
    for b in range(B):
        n = int(fb.a_num[t, b])
        embs.append(action_embedding(table[fb.vp[t, b]], fb.cand_view[t, b, :n],
                                     fb.cand_heading[t, b, :n],
                                     fb.cand_elevation[t, b, :n], loc))
    all_u, is_valid = action_variable(embs)
    return X, all_u, is_valid


def dense_speaker_inputs(sb, table, loc_table):
    """speaker.py:68-121 _batch_observations_and_actions over a synth.SpeakerBatch:
    per path step [B,F] chosen-action embeddings and [B,V,F] panoramas (zeros on
    padded steps), and the path mask (True = padded)."""
    Tp, B = sb.vp.shape
    F_ = table.shape[-1] + loc_table.shape[-1]
    acts = [np.zeros((B, F_), np.float32) for _ in range(Tp)]
    feats = [np.zeros((B, table.shape[1], F_), np.float32) for _ in range(Tp)]
    mask = np.ones((B, Tp), bool)
    for b in range(B):
        n = int(sb.path_len[b])
        mask[b, :n] = False
        for s in range(n):
            feats[s][b] = panorama_feature(table[sb.vp[s, b]], sb.view[s, b], loc_table)
            if not sb.act_is_stop[s, b]:
                emb = action_embedding(
                    table[sb.vp[s, b]], [0, sb.act_view[s, b]], [0.0, sb.act_heading[s, b]],
                    [0.0, sb.act_elevation[s, b]])
                acts[s][b] = emb[1]
    return acts, feats, mask


# ------------------------------------------------------------------------------------------------
# tasks/R2R/eval.py:56-139 (Evaluation._get_nearest / _score_item / score_results): navigation error =
# shortest-path distance from the LAST viewpoint of a trajectory to the goal, oracle error = from the
# closest viewpoint on it, success = error < 3 m; the summary averages over all scored instructions.
# `graphs`: scan -> object with .distance(a, b) (speaker_follower_amd.env.NavGraph); `gt`: path_id ->
# item with 'scan' and 'path'.
# ------------------------------------------------------------------------------------------------
def score_results(gt, graphs, results, error_margin=3.0):
    per_item, acc = {}, dict(nav_error=[], oracle_error=[], steps=[], lengths=[], success=[], oracle_success=[])
    for instr_id, res in results.items():
        path = res['trajectory']
        item = gt[int(instr_id.split('_')[0])]
        g = graphs[item['scan']]
        assert item['path'][0] == path[0][0], 'Result trajectories should include the start position'
        goal = item['path'][-1]
        nav_error = g.distance(path[-1][0], goal)                                          # :66
        oracle_error = min(g.distance(p[0], goal) for p in path)                           # :48-54, :67
        length = sum(g.distance(a[0], b[0]) for a, b in zip(path[:-1], path[1:]))          # :69-73
        per_item[instr_id] = dict(nav_error=nav_error, oracle_error=oracle_error, steps=len(path) - 1,
                                  length=length, success=nav_error < error_margin,
                                  oracle_success=oracle_error < error_margin)
        for k, v in (('nav_error', nav_error), ('oracle_error', oracle_error), ('steps', len(path) - 1),
                     ('lengths', length), ('success', nav_error < error_margin),
                     ('oracle_success', oracle_error < error_margin)):
            acc[k].append(v)
    n = len(results)
    summary = dict(nav_error=float(np.mean(acc['nav_error'])), oracle_error=float(np.mean(acc['oracle_error'])),
                   steps=float(np.mean(acc['steps'])), lengths=float(np.mean(acc['lengths'])),
                   success_rate=float(sum(acc['success'])) / n, oracle_rate=float(sum(acc['oracle_success'])) / n)
    return summary, per_item
