"""A simulator-free environment with the reference's R2RBatch interface (env.py:667-854:
reset / observe / step / reset_epoch / gold_obs_actions_and_instructions) that replays a
synth.FollowerBatch / synth.SpeakerBatch and hands out the reference's observation dictionaries
(env.py:775-795).  It exists so that Seq2SeqAgent / Seq2SeqSpeaker can be exercised end to end
where the Matterport simulator cannot be built; observations are assembled on the host with numpy
exactly like env.py:60-75 and :771-774 do."""
import numpy as np

from .features import build_loc_table


def _action_embedding(features, cand_view, cand_heading, cand_elevation, loc=128):
    """env.py:60-75 (stop action = zero row; sin/cos evaluated in float64)."""
    n, img, g = len(cand_view), features.shape[-1], loc // 4
    emb = np.zeros((n, img + loc), np.float32)
    for a in range(1, n):
        emb[a, :img] = features[cand_view[a]]
        h, e = np.float64(cand_heading[a]), np.float64(cand_elevation[a])
        emb[a, img:img + g] = np.sin(h)
        emb[a, img + g:img + 2 * g] = np.cos(h)
        emb[a, img + 2 * g:img + 3 * g] = np.sin(e)
        emb[a, img + 3 * g:] = np.cos(e)
    return emb


class SyntheticR2REnv:
    def __init__(self, fb, table, loc=128):
        self.fb, self.table = fb, table
        self.loc_table = build_loc_table(table.shape[1], loc)
        self.batch_size = fb.vp.shape[1]
        self.beam_size = 1
        self.t = 0
        self.image_features_list = [None]

    def reset_epoch(self):
        self.t = 0

    def reset(self, sort=False, beamed=False, load_next_minibatch=True):
        self.t = 0
        return list(range(self.batch_size))

    def step(self, world_states, actions, last_obs, beamed=False):
        self.t = min(self.t + 1, self.fb.vp.shape[0] - 1)
        return world_states

    def _ob(self, t, b):
        fb = self.fb
        n = int(fb.a_num[t, b])
        feats = self.table[fb.vp[t, b]]
        teacher = int(fb.target[t, b])
        return {
            'instr_id': 'synth_%d' % b, 'scan': 'synth', 'viewpoint': str(int(fb.vp[t, b])),
            'viewIndex': int(fb.view[t, b]), 'heading': 0.0, 'elevation': 0.0, 'step': t,
            'feature': [np.concatenate((feats, self.loc_table[fb.view[t, b]]), axis=-1)],   # env.py:773
            'adj_loc_list': [dict(absViewIndex=int(fb.cand_view[t, b, a]),
                                  rel_heading=float(fb.cand_heading[t, b, a]),
                                  rel_elevation=float(fb.cand_elevation[t, b, a])) for a in range(n)],
            'action_embedding': _action_embedding(feats, fb.cand_view[t, b, :n],
                                                  fb.cand_heading[t, b, :n],
                                                  fb.cand_elevation[t, b, :n]),
            'teacher': teacher if teacher >= 0 else 0,
            'instr_encoding': fb.instr[b], 'instructions': '',
        }

    def observe(self, world_states, beamed=False, include_teacher=True):
        return [self._ob(self.t, b) for b in world_states]

    def gold_obs_actions_and_instructions(self, max_steps, load_next_minibatch=True):
        """env.py:850-854 over the synthetic teacher: per sample the observations along the teacher
        path (one more than actions) and the teacher actions up to and including stop."""
        fb = self.fb
        T, B = fb.target.shape
        path_obs, path_actions = [], []
        for b in range(B):
            obs, acts = [self._ob(0, b)], []
            for t in range(min(T, max_steps)):
                a = int(fb.target[t, b])
                if a < 0:
                    break
                acts.append(a)
                obs.append(self._ob(min(t + 1, T - 1), b))
                if a == 0:
                    break
            path_obs.append(obs)
            path_actions.append(acts)
        return path_obs, path_actions, [fb.instr[b] for b in range(B)]
