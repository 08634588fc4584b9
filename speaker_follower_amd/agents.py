"""Seq2SeqAgent / Seq2SeqSpeaker with the reference's interface (follower.py:261-1035,
speaker.py:34-410) on top of the HIP modules, for environments that hand out the reference's
observation dictionaries (env.py:775-795: 'feature', 'action_embedding', 'adj_loc_list',
'teacher', 'instr_encoding', ...).  The per-step arithmetic -- decoder step, masking,
cross-entropy, action choice, u_prev gather -- runs through the C ABI; the loop, the trajectory
book-keeping and the simulator calls stay in Python exactly where the reference has them.

When observations are available in index form, `FollowerEngine` / `SpeakerEngine` are the fast
path; this module is the drop-in path.  Beam search, state-factored search and the speaker's beam search
(follower.py:541-980, speaker.py:211-318) live in search.py and run on index-form observations.
"""
import ctypes as C
import json
import random
import sys

import numpy as np
import torch

from . import _lib
from ._lib import call
from .lazydict import LazyDict
from .follower import batch_instructions_from_encoded, FEEDBACK, PAD, EOS, BOS
from .runtime import ptr, stream, require_gpu, cands_dense, PersistentLaunchFault, gc_paused, WeightsMoved

byref = C.byref


def path_element_from_observation(ob):
    return (ob['viewpoint'], ob['heading'], ob['elevation'])           # follower.py utility


class _FollowerGlueFn(torch.autograd.Function):
    """follower.py:476-505 on dense tensors: returns (ce_term [B], live [B]); a_t, per-step scores
    and u_next are side outputs.  backward: softmax - onehot through sf_follower_glue_bwd."""

    @staticmethod
    def forward(ctx, logit, all_u, is_valid, target, feedback, ended_dev, sample_cfg):
        B, A = logit.shape
        dev = logit.device
        masked = logit.detach().clone()
        new = lambda *s, dt=torch.float32: torch.empty(*s, device=dev, dtype=dt)  # noqa: E731
        a_t, tused = new(B, dt=torch.int64), new(B, dt=torch.int64)
        score, ce, live = new(B), new(B), new(B)
        u_next = new(B, all_u.shape[2])
        cnd = cands_dense(all_u)
        g = _lib.FollowerGlue(is_valid.data_ptr(), target.data_ptr(), feedback, ended_dev.data_ptr(),
                              a_t.data_ptr(), tused.data_ptr(), score.data_ptr(), u_next.data_ptr(),
                              all_u.shape[2], None, 0, ce.data_ptr(), live.data_ptr(),
                              sample_cfg[0], sample_cfg[1], 0)
        call('sf_follower_glue_fwd', byref(cnd), B, ptr(masked), byref(g), stream())
        ctx.save_for_backward(masked, tused)
        ctx.mark_non_differentiable(live, a_t, score, u_next, masked)
        return ce, live, a_t, score, u_next, masked

    @staticmethod
    def backward(ctx, dce, *_):
        masked, tused = ctx.saved_tensors
        B, A = masked.shape
        one = torch.ones(1, device=masked.device)
        dlogit = torch.empty_like(masked)
        call('sf_follower_glue_bwd', B, A, ptr(masked), ptr(tused), ptr(one), ptr(dlogit), stream())
        return dlogit * dce.reshape(B, 1), None, None, None, None, None, None


class BaseAgent(object):
    """follower.py:107-192."""

    def __init__(self, env, results_path):
        self.env = env
        self.results_path = results_path
        random.seed(1)
        self.results = {}
        self.losses = []

    def write_results(self):
        results = {k: {'instr_id': v['instr_id'], 'trajectory': v['trajectory']}
                   for k, v in self.results.items()}
        with open(self.results_path, 'w') as f:
            json.dump(results, f)

    def rollout(self):
        raise NotImplementedError

    def test(self):
        self.env.reset_epoch()
        self.losses = []
        self.results = {}
        looped = False
        while True:                                                    # follower.py:145-188
            for result in self.rollout():
                if result['instr_id'] in self.results:
                    looped = True
                else:
                    self.results[result['instr_id']] = result
            if looped:
                break
        return self.results


class Seq2SeqAgent(BaseAgent):
    """follower.py:261-1035 (greedy / teacher / sample rollouts, scoring, train, save/load)."""
    feedback_options = ['teacher', 'argmax', 'sample']

    def __init__(self, env, results_path, encoder, decoder, episode_len=10, beam_size=1,
                 reverse_instruction=True, max_instruction_length=80):
        super().__init__(env, results_path)
        self.encoder, self.decoder = encoder, decoder
        self.episode_len = episode_len
        self.losses = []
        self.beam_size = beam_size
        self.reverse_instruction = reverse_instruction
        self.max_instruction_length = max_instruction_length
        self.feedback = 'argmax'
        self.loss = 0
        self._sample_seed = torch.initial_seed() & 0xFFFFFFFF
        self._sample_count = 0
        self.store = None          # features.FeatureStore: enables the index-form search procedures
        self.nav_table = None      # nav.NavTable: rollouts step the environment ON THE DEVICE
        self._engine = None
        # An environment that carries its feature store and no host copy of the table (compat.env.R2RBatch, what
        # the reference's train.py builds through the bare-name shim) can only be walked on the device: rollouts
        # then use nav.table_for(env, store) of whatever env is current (train.py switches agent.env between
        # the training and the validation environments, train.py:97, 111).
        self.auto_device_env = True

    def use_device_env(self, nav_table):
        """Opt in to device-resident navigation (nav.py): `_rollout_with_loss` -- and with it `train`,
        `test` and `rollout` -- then runs the whole episode with one host sync instead of a D2H copy
        plus Python env.step / observe per step.  Needs `self.store`; the env must be the
        R2RIndexEnv the table was built from.  The result dictionaries carry instr_id / trajectory /
        actions / scores (no per-step 'observations': nothing observes on the host any more)."""
        from .follower import FollowerEngine
        if self.store is None:
            raise RuntimeError('use_device_env needs a features.FeatureStore (agent.store)')
        self.nav_table = nav_table
        self._engine = FollowerEngine(self.encoder, self.decoder, self.store)
        self._engine.dropout_seed = self._sample_seed ^ 0x1B873593

    def _env_store(self):
        """The feature store the current env was built over (compat.env.MeanPooledImageFeatures.store), if any."""
        feats = getattr(self.env, 'image_features_list', None) or [None]
        return getattr(feats[0], 'store', None)

    def _device_table(self):
        """The navigation table to roll out on, or None for the per-step host loop: the one given to
        use_device_env, else (auto) the current env's own when it can only be walked on the device."""
        if self.nav_table is not None:
            return self.nav_table
        if not self.auto_device_env or getattr(self.env, 'host_table', 1) is not None or not hasattr(self.env, 'graphs'):
            return None
        if self.store is None:
            self.store = self._env_store()
        if self.store is None:
            return None
        from .follower import FollowerEngine
        from .nav import table_for
        if self._engine is None:
            self._engine = FollowerEngine(self.encoder, self.decoder, self.store)
            self._engine.dropout_seed = self._sample_seed ^ 0x1B873593
        return table_for(self.env, self.store)

    def _rollout_on_device(self, table=None, reissue=False):
        """One rollout on the device-resident environment.  `FollowerEngine.run` checks the fault word of the
        persistent encoder launch at its sync and re-issues the rollout on the per-step kernels by itself;
        reissue=True (train(): a fault raised by the BACKWARD) replays the current minibatch that way."""
        from .nav import DeviceNavBatch
        if not reissue:
            self.env.reset(sort=True)
        nav = table if table is not None else self.nav_table
        items = list(self.env.batch)
        # what the previous rollout prepared while its kernels ran (the same items, or it is not used)
        ahead, batch = self.__dict__.pop('_rollout_ahead', None), None
        if not (ahead is not None and ahead[0] is nav and len(ahead[1]) == len(items)
                and all(a is b for a, b in zip(ahead[1], items))):
            ahead = None
        if (self.test_graph and not reissue and not self.decoder.training and self.feedback == 'argmax'
                and self._engine.group is None):
            return self._rollout_on_graph(nav, items, ahead[2] if ahead is not None and isinstance(ahead[2], dict) else None)
        if ahead is not None and isinstance(ahead[2], DeviceNavBatch) and ahead[2].steps == self.episode_len:
            batch = ahead[2]
        if batch is None:
            batch = DeviceNavBatch(nav, items, self.episode_len, max_length=self.max_instruction_length,
                                   reverse=self.reverse_instruction)

        def prepare_next():
            # the minibatch the next env.reset will draw (peeked, not drawn: nothing changes if nobody comes for it):
            # encoded and packed on the host while this rollout's kernels run; its device tensors are filled behind them
            peek = getattr(self.env, 'peek_next_minibatch', None)
            nxt = peek(True) if peek is not None and self.prepare_ahead else None
            if nxt is not None:
                self._rollout_ahead = (nav, nxt, DeviceNavBatch(nav, nxt, self.episode_len,
                                                                max_length=self.max_instruction_length,
                                                                reverse=self.reverse_instruction))
        keep = getattr(self.encoder, 'persistent', True)
        if reissue:
            self.encoder.persistent = False
        try:
            st = self._engine.run(batch, self.episode_len, self.feedback, train=self.decoder.training,
                                  while_running=None if reissue else prepare_next)
        finally:
            self.encoder.persistent = keep
        self.loss = st.loss
        traj = batch.trajectories(st)                   # the one host sync of the rollout
        for tr, it in zip(traj, items):
            tr['instr_encoding'] = it['instr_encoding']
        self.losses.append(float(st.loss.detach()))
        return traj

    test_graph = True            # argmax inference rollouts on the device environment: one hipGraph replay per minibatch

    def _rollout_on_graph(self, nav, items, host=None):
        """An inference rollout (agent.test, follower.py:987-999) as ONE graph replay over a fixed-shape minibatch: the
        items are written into the captured batch's tensors (one pinned copy), the next minibatch is peeked and encoded
        while the replay runs, the fault word is read where the results are.  The results (and the fault words) come
        down asynchronously, and the NEXT minibatch's replay is issued right behind that download: the device runs it
        while the host builds this minibatch's dictionaries; the next call takes it over only for exactly the peeked
        items under exactly the current weight versions.  A starved persistent launch re-issues the minibatch on the
        per-step kernels (and drops what was issued ahead); weights that MOVED since the capture (load_state_dict into
        new tensors) mean a new capture.  Instructions are padded to max_instruction_length (the eager rollout pads to
        the minibatch's longest: equal up to the summation order of the padded attention columns).  `self.loss` is a
        host tensor on this path."""
        from .nav import DeviceNavBatch
        from .runtime import take_fault, fault_views
        eng, dev = self._engine, self._device()
        key = (id(nav), id(eng), self.episode_len, len(items), self.max_instruction_length, self.reverse_instruction)
        graphs = self.__dict__.setdefault('_test_graphs', {})  # (train.py alternates between its validation environments)
        S = self.episode_len
        # (key, items, weight versions): the replay the previous call issued ahead -- good for exactly these items under
        # exactly these weights
        versions = tuple((p_.data_ptr(), p_._version) for m in (self.encoder, self.decoder) for p_ in m.parameters())
        flying = self.__dict__.pop('_rollout_inflight', None)
        hit = (flying is not None and flying[0] == key and key in graphs and flying[2] == versions
               and len(flying[1]) == len(items) and all(a is b for a, b in zip(flying[1], items)))
        if hit:
            replay, st, batch, _, pinned = graphs[key]         # already loaded and replayed: its results wait on the device
        for attempt in () if hit else (0, 1):
            cached = graphs.get(key)
            if cached is None:
                batch = DeviceNavBatch(nav, items, S, max_length=self.max_instruction_length,
                                       reverse=self.reverse_instruction, fixed_shapes=True, host=host)
                replay, st = eng.capture(batch, S, 'argmax')
                pinned = [torch.empty(t.shape, dtype=t.dtype).pin_memory()
                          for t in (batch.row[:S + 1], batch.view[:S + 1], st.actions, st.step_scores, st.loss_buf)]
                while len(graphs) >= 3:
                    graphs.pop(next(iter(graphs)))
                graphs[key] = (replay, st, batch, nav, pinned)  # (nav kept alive: the key holds its id)
            else:
                replay, st, batch, _, pinned = cached
                batch.load(items, host)
            take_fault(dev)                                   # (whatever an earlier pass left behind is not ours)
            try:
                replay()
                break
            except WeightsMoved:                              # (only this: a device error is not a reason to recapture)
                if attempt:
                    raise
                graphs.pop(key, None)                         # a weight moved: capture again
        # this minibatch's results leave the device buffers before anything else is written into them ...
        for dst, src in zip(pinned, (batch.row[:S + 1], batch.view[:S + 1], st.actions, st.step_scores, st.loss_buf)):
            dst.copy_(src, non_blocking=True)
        # (the fault words travel with them: runtime.take_fault's synchronous copy would wait for the replay issued below)
        faults = fault_views(dev)
        fpin = self.__dict__.get('_fault_pin')
        if fpin is None or fpin.numel() != len(faults):
            fpin = self._fault_pin = torch.empty(len(faults), dtype=torch.int32).pin_memory()
        fpin.copy_(torch.cat(faults) if len(faults) > 1 else faults[0], non_blocking=True)
        done = torch.cuda.Event()
        done.record()
        # ... and the NEXT minibatch (peeked, not drawn) is encoded, loaded and replayed right behind them: the device
        # runs it while the host builds this minibatch's dictionaries
        peek = getattr(self.env, 'peek_next_minibatch', None)
        nxt = peek(True) if peek is not None and self.prepare_ahead else None
        if nxt is not None and len(nxt) == len(items):
            nhost = DeviceNavBatch.host_arrays_for(nav, nxt, self.max_instruction_length, self.reverse_instruction, True)
            try:
                batch.load(nxt, nhost)
                replay()
                self._rollout_inflight = (key, nxt, versions)
            except WeightsMoved:
                self._rollout_ahead = (nav, nxt, nhost)
        done.synchronize()
        rows, views, acts, sc, loss = (p_.numpy().copy() for p_ in pinned)
        if any(fpin.tolist()):
            self.__dict__.pop('_rollout_inflight', None)      # (what was issued ahead is not trusted either: reloaded next)
            torch.cuda.synchronize(dev)
            take_fault(dev)                                   # (read and cleared)
            batch.load(items)
            keep = getattr(self.encoder, 'persistent', True)
            self.encoder.persistent = False
            try:
                with torch.no_grad():
                    st2 = eng.rollout(batch, S, 'argmax', train=False)
            finally:
                self.encoder.persistent = keep
            if take_fault(dev):
                raise PersistentLaunchFault('the per-step re-issue of an inference rollout raised a fault again')
            eng.fallbacks += 1
            rows, views = batch.row[:S + 1].cpu().numpy(), batch.view[:S + 1].cpu().numpy()
            acts, sc, loss = st2.actions.cpu().numpy(), st2.step_scores.cpu().numpy(), st2.loss_buf.cpu().numpy()
        self.loss = torch.tensor(float(loss.reshape(-1)[0]))        # (a host tensor: a copy to the device would queue behind the replay issued ahead)
        traj = batch.trajectories_from(items, S, rows, views, acts, sc)
        for tr, it in zip(traj, items):
            tr['instr_encoding'] = it['instr_encoding']
        self.losses.append(float(loss.reshape(-1)[0]))
        return traj

    # ---- tensor assembly (follower.py:291-332): numpy stacks -> device tensors
    def _device(self):
        return next(self.decoder.parameters()).device

    def _feature_variables(self, obs):
        feats = np.stack([ob['feature'][0] for ob in obs])
        return [torch.from_numpy(feats).to(self._device())]

    def _action_variable(self, obs):
        max_a = max(len(ob['adj_loc_list']) for ob in obs)
        dim = obs[0]['action_embedding'].shape[-1]
        is_valid = np.zeros((len(obs), max_a), np.float32)
        emb = np.zeros((len(obs), max_a, dim), np.float32)
        for i, ob in enumerate(obs):
            n = len(ob['adj_loc_list'])
            is_valid[i, :n] = 1.0
            emb[i, :n] = ob['action_embedding']
        dev = self._device()
        return torch.from_numpy(emb).to(dev), torch.from_numpy(is_valid).to(dev), is_valid

    def _teacher_action(self, obs, ended):
        a = torch.tensor([ob['teacher'] if not ended[i] else -1 for i, ob in enumerate(obs)],
                         dtype=torch.int64)
        return a.to(self._device())

    def _proc_batch(self, obs):
        enc = [ob['instr_encoding'] for ob in obs]
        return batch_instructions_from_encoded(enc, self.max_instruction_length,
                                               reverse=self.reverse_instruction,
                                               device=self._device())

    def rollout(self):
        if self.beam_size == 1:                                        # follower.py:334-340
            return self._rollout_with_loss()
        assert self.beam_size >= 1
        beams, _, _ = self.beam_search(self.beam_size)
        return [beam[0] for beam in beams]

    def beam_search(self, beam_size, load_next_minibatch=True, mask_undo=False):
        """follower.py:541-718 (search.py; needs `self.store` and index-form observations)."""
        from . import search
        return search.beam_search(self, beam_size, load_next_minibatch, mask_undo)

    def state_factored_search(self, completion_size, successor_size, load_next_minibatch=True,
                              mask_undo=False, first_n_ws_key=4):
        """follower.py:720-980."""
        from . import search
        return search.state_factored_search(self, completion_size, successor_size,
                                            load_next_minibatch, mask_undo, first_n_ws_key)

    def set_beam_size(self, beam_size):
        if getattr(self.env, 'beam_size', 1) < beam_size:
            self.env.set_beam_size(beam_size)
        self.beam_size = beam_size

    def _step(self, u_prev, obs, h, c, ctx, seq_mask, target, feedback, ended_dev):
        """One decoder step + glue on dense observations (follower.py:469-505)."""
        f_t = self._feature_variables(obs)[0]
        all_u, is_valid, _ = self._action_variable(obs)
        h, c, alpha, logit, alpha_v = self.decoder(u_prev, all_u, f_t, h, c, ctx, seq_mask)
        self._sample_count += 1
        ce, live, a_t, score, u_next, masked = _FollowerGlueFn.apply(
            logit, all_u, is_valid, target, FEEDBACK[feedback], ended_dev,
            (self._sample_seed, self._sample_count))
        n_live = live.sum()
        loss_t = ce.sum() / n_live.clamp(min=1.0)                    # CrossEntropyLoss mean over live
        return h, c, loss_t, a_t, score, u_next

    def _rollout_with_loss(self, reissue=False):
        """follower.py:430-539."""
        table = self._device_table()
        if table is not None:
            return self._rollout_on_device(table, reissue)
        world_states = self.env.reset(sort=True)
        obs = np.array(self.env.observe(world_states))
        B = len(obs)
        seq, seq_mask, seq_lengths = self._proc_batch(obs)
        self.loss = 0
        feedback = self.feedback
        ctx, h, c = self.encoder(seq, seq_lengths)
        traj = [{'instr_id': ob['instr_id'], 'trajectory': [path_element_from_observation(ob)],
                 'actions': [], 'scores': [], 'observations': [ob],
                 'instr_encoding': ob['instr_encoding']} for ob in obs]
        dev = self._device()
        u_prev = self.decoder.u_begin.to(dev).expand(B, -1).contiguous()
        ended = np.array([False] * B)
        ended_dev = torch.zeros(B, dtype=torch.uint8, device=dev)
        seq_scores = torch.zeros(B, device=dev)
        for t in range(self.episode_len):
            target = self._teacher_action(obs, ended)
            ended_dev.zero_()      # `target` already carries -1 for ended rows (follower.py:327)
            h, c, loss_t, a_t, score, u_prev = self._step(u_prev, obs, h, c, ctx, seq_mask, target,
                                                          feedback, ended_dev)
            self.loss = self.loss + loss_t
            seq_scores = seq_scores + score
            actions = a_t.tolist()                                     # the per-step D2H sync (:509-511)
            world_states = self.env.step(world_states, actions, obs)
            obs = self.env.observe(world_states)
            sc, ss = score.tolist(), seq_scores.tolist()
            for i, ob in enumerate(obs):
                if not ended[i]:
                    traj[i]['trajectory'].append(path_element_from_observation(ob))
                    traj[i]['score'] = ss[i]
                    traj[i]['scores'].append(sc[i])
                    traj[i]['actions'].append(actions[i])
                    traj[i]['observations'].append(ob)
            for i in range(B):
                if actions[i] == 0:
                    ended[i] = True
            if ended.all():
                break
        self.losses.append(float(self.loss.detach()) if torch.is_tensor(self.loss) else float(self.loss))
        return traj

    def _score_obs_actions_and_instructions(self, path_obs, path_actions, encoded_instructions):
        """follower.py:342-428: teacher-forced scoring of given paths."""
        B = len(path_obs)
        assert len(path_actions) == B and len(encoded_instructions) == B
        dev = self._device()
        seq, seq_mask, seq_lengths, perm = batch_instructions_from_encoded(
            encoded_instructions, self.max_instruction_length, reverse=self.reverse_instruction,
            sort=True, device=dev)
        loss = 0
        ctx, h, c = self.encoder(seq, seq_lengths)
        u_prev = self.decoder.u_begin.to(dev).expand(B, -1).contiguous()
        ended = np.array([False] * B)
        ended_dev = torch.zeros(B, dtype=torch.uint8, device=dev)
        seq_scores = torch.zeros(B, device=dev)
        traj = [{'instr_id': po[0]['instr_id'], 'trajectory': [path_element_from_observation(po[0])],
                 'actions': [], 'scores': [], 'observations': [po[0]],
                 'instr_encoding': po[0]['instr_encoding']} for po in path_obs]
        obs = None
        for t in range(self.episode_len):
            nxt, tgt = [], []
            for pi, si in enumerate(perm):
                if t < len(path_actions[si]):
                    tgt.append(path_actions[si][t])
                    nxt.append(path_obs[si][t])
                else:
                    tgt.append(-1)
                    nxt.append(obs[pi])
            obs = nxt
            target = torch.tensor(tgt, dtype=torch.int64, device=dev)
            h, c, loss_t, a_t, score, u_prev = self._step(u_prev, obs, h, c, ctx, seq_mask, target,
                                                          'teacher', ended_dev)
            ended_dev.zero_()
            loss = loss + loss_t
            live = (target != -1).to(score.dtype)
            seq_scores = seq_scores + score * live                     # :405 ignore_index rows score 0
            acts, sc, ss = a_t.tolist(), (score * live).tolist(), seq_scores.tolist()
            for pi, si in enumerate(perm):
                if not ended[pi]:
                    traj[si]['trajectory'].append(path_element_from_observation(obs[pi]))
                    traj[si]['score'] = ss[pi]
                    traj[si]['scores'].append(sc[pi])
                    traj[si]['actions'].append(acts[pi])
            for i in range(B):
                if acts[i] == 0:
                    ended[i] = True
            if ended.all():
                break
        return traj, loss

    def test(self, use_dropout=False, feedback='argmax', allow_cheat=False, beam_size=1):
        """follower.py:987-999."""
        if not allow_cheat:
            assert feedback in ['argmax', 'sample']
        self.feedback = feedback
        for m in (self.encoder, self.decoder):
            m.train() if use_dropout else m.eval()
        self.set_beam_size(beam_size)
        self.__dict__.pop('_rollout_inflight', None)          # (nothing issued ahead by an earlier loop is ours)
        # (test() reads results, nothing differentiates them: without autograd a rollout keeps no tape -- a differentiable
        # state lives until the cyclic collector finds it, a few hundred MB per minibatch)
        with torch.no_grad():
            return super().test()

    def train(self, encoder_optimizer, decoder_optimizer, n_iters, feedback='teacher'):
        """follower.py:1001-1020."""
        assert all(f in self.feedback_options for f in feedback.split('+'))
        self.feedback = feedback
        self.encoder.train()
        self.decoder.train()
        self.losses = []
        if self._graph_trainable(encoder_optimizer, decoder_optimizer):
            return self._train_on_graph(encoder_optimizer, decoder_optimizer, n_iters)
        for _ in range(1, n_iters + 1):
            encoder_optimizer.zero_grad()
            decoder_optimizer.zero_grad()
            self._rollout_with_loss()
            self.loss.backward()
            # a starved persistent launch of the BACKWARD has poisoned the gradients: never step on them -- replay
            # the minibatch on the per-step kernels (the forward's own check sits inside FollowerEngine.run)
            if _persistent_fault(self._device()):
                encoder_optimizer.zero_grad()
                decoder_optimizer.zero_grad()
                self.losses.pop()
                keep = getattr(self.encoder, 'persistent', True)
                self.encoder.persistent = False
                try:
                    # (the device environment replays the SAME minibatch; the per-step host loop cannot rewind its
                    # simulators and trains on the next one)
                    self._rollout_with_loss(reissue=True)
                    self.loss.backward()
                finally:
                    self.encoder.persistent = keep
                if _persistent_fault(self._device()):
                    raise PersistentLaunchFault('the per-step re-issue of a training iteration raised a fault again')
            encoder_optimizer.step()
            decoder_optimizer.step()

    # ---- training iterations as hipGraph replays (runtime.TrainingGraph) ------------------------------------------
    train_graph = True           # False: every iteration issued launch by launch (the loop above)

    def _graph_trainable(self, encoder_optimizer, decoder_optimizer):
        """Whole iterations can be replayed when the environment is walked on the device, both optimizers keep their
        state in flat buffers with device-side step counters (optim.FusedAdam) and nothing leaves the process."""
        from .optim import FusedAdam
        if not self.train_graph or '+' in self.feedback or self.beam_size != 1:
            return False
        for o in (encoder_optimizer, decoder_optimizer):
            if not (isinstance(o, FusedAdam) or (self.adopt_torch_adam and FusedAdam.adoptable(o))):
                return False
        if self._device_table() is None or self._engine is None:
            return False
        eng = self._engine
        # (a process group: iterations replay as SEGMENTS cut at the collective points, FollowerEngine._capture_training_dp;
        # that needs the gradient buckets the all-reduces work on)
        dp_ok = (eng.group is None and eng.grad_sync is None) or (eng.group is not None and eng.grad_sync is not None)
        return (dp_ok and getattr(self.encoder, 'num_directions', 1) == 1
                and getattr(self.encoder, 'persistent', True))

    def _train_on_graph(self, encoder_optimizer, decoder_optimizer, n_iters):
        """follower.py:1001-1020 with every iteration after the first ONE graph replay: the next minibatch is written
        into the captured batch's tensors (nav.DeviceNavBatch.load), dropout sites / samples / Adam steps advance
        through device words, and the optimizer steps are GUARDED by the fault word of the persistent launches
        (sf_adam_step_dev: a starved launch turns them into no-ops) -- the host looks at the word where it reads the
        loss and re-issues that iteration on the per-step kernels.  Same sites, same numbers as the eager loop
        (tests/test_gpu_agents.py)."""
        from .nav import DeviceNavBatch
        from .runtime import take_fault
        table, eng, dev = self._device_table(), self._engine, self._device()
        given = (encoder_optimizer, decoder_optimizer)
        opts = tuple(self._fused_for(o) for o in given)
        try:
            return self._replay_iterations(opts, n_iters, table, eng, dev)
        finally:
            for o, f in zip(given, opts):             # the reference's own optimizer objects get their state back
                if f is not o:
                    f.store_torch_state(o)

    prepare_ahead = True         # train() on graphs: the next minibatch's host work under the current replay
    adopt_torch_adam = True      # train(): a plain torch.optim.Adam (train.py:263-268) is mirrored by an optim.FusedAdam

    def _fused_for(self, opt):
        """`opt` itself (optim.FusedAdam), or the FusedAdam that mirrors this torch.optim.Adam: same parameters (their
        data moves into one flat buffer, once), hyper-parameters and state taken over at the start of every train() call
        and handed back at its end -- checkpoints of `opt.state_dict()` and steps taken elsewhere stay consistent."""
        import weakref
        from .optim import FusedAdam
        if isinstance(opt, FusedAdam):
            return opt
        mirrors = self.__dict__.setdefault('_fused_mirrors', {})
        hit = mirrors.get(id(opt))
        if hit is None or hit[0]() is not opt:
            g = opt.param_groups[0]
            fused = FusedAdam(g['params'], lr=g['lr'], betas=tuple(g['betas']), eps=g['eps'], weight_decay=g['weight_decay'])
            mirrors[id(opt)] = hit = (weakref.ref(opt), fused)
        hit[1].load_torch_state(opt)
        return hit[1]

    def _replay_iterations(self, opts, n_iters, table, eng, dev):
        from .nav import DeviceNavBatch
        from .runtime import take_fault
        encoder_optimizer, decoder_optimizer = opts
        # (a captured iteration holds the optimizers' hyper-parameters as kernel arguments: a changed learning rate is a
        # new graph)
        hyper = tuple((g['lr'], tuple(g['betas']), g['eps'], g['weight_decay']) for o in opts for g in o.param_groups)
        key = (id(encoder_optimizer), id(decoder_optimizer), self.feedback, id(table), self.episode_len, hyper)
        cached = self.__dict__.get('_train_graph_state')
        take_fault(dev)                                       # (whatever an earlier pass left behind is not ours)
        ahead = None            # the NEXT minibatch and its host arrays, formed while the device ran this iteration
        for it in range(n_iters):
            if (self.pipeline_replays and self.prepare_ahead and cached is not None and cached[0] == key
                    and cached[2].batch_size == self.env.batch_size and eng.group is None):
                # the graph exists: the remaining iterations with one replay always queued behind the running one
                return self._replay_pipelined(opts, n_iters - it, cached, eng, dev, ahead)
            if ahead is None:
                self.env.reset(sort=True)
                items, host = list(self.env.batch), None
            else:
                items, host = ahead
                ahead = None
            before = [o.host_steps() for o in opts]
            where = (eng.site_next, eng.iteration)            # the dropout / sample sites this iteration starts at
            if cached is not None and (cached[0] != key or cached[2].batch_size != len(items)):
                cached = None
            if cached is None:
                batch = DeviceNavBatch(table, items, self.episode_len, max_length=self.max_instruction_length,
                                       reverse=self.reverse_instruction, fixed_shapes=True)
                for o in opts:
                    o.guard_faults = True
                tg = eng.capture_training(batch, self.episode_len, self.feedback, optimizers=opts, zero=eng.grad_sync)
                cached = self._train_graph_state = (key, tg, batch)
                st = tg.first                                 # (the capture ran this iteration eagerly)
            else:
                _, tg, batch = cached
                batch.load(items, host)
                st = tg.replay()
            if self.prepare_ahead and it + 1 < n_iters:
                # the replay is in flight: draw the next minibatch (the same env.reset calls in the same order, earlier)
                # and encode it now; its copies wait until this iteration's loss has been read
                self.env.reset(sort=True)
                nxt = list(self.env.batch)
                ahead = (nxt, batch._host_arrays(nxt) if len(nxt) == batch.batch_size else None)
            loss = float(st.loss_buf)                         # the iteration's one host sync
            bits = take_fault(dev)
            if eng.group is not None:
                # every rank takes the same decision (MAX: RCCL has no bitwise reductions): a re-issue launches
                # collectives, and those only match if all ranks re-issue
                t_ = torch.tensor([bits], device=dev, dtype=torch.int32)
                torch.distributed.all_reduce(t_, op=torch.distributed.ReduceOp.MAX, group=eng.group)
                bits = int(t_.item())
            if bits:
                # a persistent launch starved: the guarded optimizer steps did nothing.  The same minibatch again on
                # the per-step kernels, launch by launch
                if eng.grad_sync is not None:
                    eng.grad_sync.abort()                     # (nothing of the faulted round is in flight: replays wait)
                for o, b in zip(opts, before):
                    o.set_host_steps(b)
                    o.zero_grad()
                eng.site_next, eng.iteration = where          # (the re-issue draws the masks / samples the replay drew)
                keep = getattr(self.encoder, 'persistent', True)
                self.encoder.persistent = False
                try:
                    st = eng.rollout(batch, self.episode_len, self.feedback, train=True)
                    st.loss.backward()
                finally:
                    self.encoder.persistent = keep
                if eng.grad_sync is not None:
                    eng.grad_sync.wait()
                loss = float(st.loss_buf)
                if take_fault(dev):
                    raise PersistentLaunchFault('the per-step re-issue of a training iteration raised a fault again')
                eng.fallbacks += 1
                for o in opts:
                    o.step()
            self.loss = st.loss_buf.reshape(())
            self.losses.append(loss)

    pipeline_replays = True      # train() on graphs: replay i + 1 is queued before the loss of replay i is read

    def _replay_pipelined(self, opts, n, cached, eng, dev, first=None):
        """`n` iterations of `_replay_iterations` with the device never waiting for the host: replay i + 1 (its minibatch
        loaded behind replay i on the same stream) is queued BEFORE the host waits for iteration i, whose loss and fault
        words come down asynchronously into pinned memory.  Nothing on the device depends on the host reading a loss;
        the one thing that does is the fault path: the guarded optimizer steps of a faulted iteration AND of the one
        queued behind it do nothing (the fault word stays raised until the host clears it), so the host re-issues the
        faulted minibatch on the per-step kernels and queues the other one again.  Same minibatches in the same order,
        same sites, same steps: losses and weights of the serial loop (tests/test_gpu_nav.py)."""
        import collections
        from .runtime import take_fault, fault_views
        _, tg, batch = cached
        def pins_for(n_words):
            # (one fault word per workspace; a workspace created later -- a side stream of the per-step fallback --
            # makes the list longer: the pinned pair is re-made to match)
            pins = self.__dict__.get('_train_pins')
            if pins is None or pins[0][1].numel() != n_words:
                pins = self._train_pins = [(torch.empty(1, dtype=torch.float32).pin_memory(),
                                            torch.empty(n_words, dtype=torch.int32).pin_memory()) for _ in range(2)]
            return pins
        todo = collections.deque()     # minibatches drawn and not issued yet: (items, host arrays or None)
        drawn = issued = 0
        if first is not None:          # (what the serial loop had already drawn for its next iteration)
            todo.append(first)
            drawn = 1
        flight = None                  # the iteration whose loss has not been read: (slot, event, steps before it, items)

        def draw():
            nonlocal drawn
            self.env.reset(sort=True)
            nxt = list(self.env.batch)
            todo.append((nxt, batch._host_arrays(nxt) if len(nxt) == batch.batch_size else None))
            drawn += 1
        while issued < n or flight is not None:
            cur = None
            if issued < n:
                if not todo:
                    draw()
                items, host = todo.popleft()
                before = ([o.host_steps() for o in opts], eng.site_next, eng.iteration)
                batch.load(items, host)
                st = tg.replay()
                slot = issued & 1
                views = fault_views(dev)
                if flight is not None and flight[4][0][1].numel() != len(views):
                    flight[1].synchronize()                   # (the pair in flight is read before it is replaced)
                pins = pins_for(len(views))
                pins[slot][0].copy_(st.loss_buf.reshape(1), non_blocking=True)
                pins[slot][1].copy_(torch.cat(views) if len(views) > 1 else views[0], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record()
                cur = (slot, ev, before, items, pins)
                issued += 1
            if flight is not None:
                slot, ev, before_f, items_f, pins = flight
                ev.synchronize()
                if any(pins[slot][1].tolist()):
                    torch.cuda.synchronize(dev)               # (the replay queued behind it did nothing either)
                    take_fault(dev)
                    for o, b in zip(opts, before_f[0]):
                        o.set_host_steps(b)
                        o.zero_grad()
                    # the same sites too: the faulted replay and the one queued behind it advanced the host mirrors
                    eng.site_next, eng.iteration = before_f[1], before_f[2]
                    batch.load(items_f)
                    keep = getattr(self.encoder, 'persistent', True)
                    self.encoder.persistent = False
                    try:
                        st2 = eng.rollout(batch, self.episode_len, self.feedback, train=True)
                        st2.loss.backward()
                    finally:
                        self.encoder.persistent = keep
                    self.losses.append(float(st2.loss_buf))
                    if take_fault(dev):
                        raise PersistentLaunchFault('the per-step re-issue of a training iteration raised a fault again')
                    eng.fallbacks += 1
                    for o in opts:
                        o.step()
                    if cur is not None:                       # queue the other one again, in front of what was drawn since
                        todo.appendleft((cur[3], None))
                        issued -= 1
                        cur = None
                else:
                    self.losses.append(float(pins[slot][0]))
            flight = cur
            if flight is not None and drawn < n and not todo:
                draw()                                        # (host work under the replay that is running)
        self.loss = tg.state.loss_buf.reshape(())

    def _encoder_and_decoder_paths(self, base_path):
        return base_path + '_enc', base_path + '_dec'

    def save(self, path):
        ep, dp = self._encoder_and_decoder_paths(path)
        torch.save(self.encoder.state_dict(), ep)
        torch.save(self.decoder.state_dict(), dp)

    def load(self, path, **kwargs):
        ep, dp = self._encoder_and_decoder_paths(path)
        self.encoder.load_state_dict(torch.load(ep, **kwargs))
        self.decoder.load_state_dict(torch.load(dp, **kwargs))


def _persistent_fault(device):
    """Fault bits of the persistent launches since the last check (runtime.take_fault); one small D2H copy."""
    from .runtime import take_fault
    return take_fault(device) if device.type == 'cuda' else 0


class _SpeakerGlueFn(torch.autograd.Function):
    """speaker.py:163-191 on a [B,vocab] logit tensor."""

    @staticmethod
    def forward(ctx, logit, target, feedback, ended_dev, pad_idx, eos_idx, sample_cfg=None):
        B, V = logit.shape
        ldv = (V + 3) & ~3
        dev = logit.device
        lg = torch.zeros(B, ldv, device=dev)
        lg[:, :V] = logit.detach()
        w_t = torch.empty(B, dtype=torch.int64, device=dev)
        score, nll, live = (torch.empty(B, device=dev) for _ in range(3))
        smp = byref(_lib.Sample(sample_cfg[0] & 0xFFFFFFFF, sample_cfg[1] & 0xFFFFFFFF, 0)) if feedback == 2 else None
        call('sf_speaker_glue_fwd', B, V, ldv, ptr(lg), ptr(target), feedback, pad_idx, eos_idx,
             ptr(ended_dev), ptr(w_t), ptr(score), ptr(nll), ptr(live), smp, stream())
        ctx.save_for_backward(lg, target)
        ctx.cfg = (B, V, ldv, pad_idx)
        ctx.mark_non_differentiable(live, w_t, score)
        return nll, live, w_t, score

    @staticmethod
    def backward(ctx, dnll, *_):
        lg, target = ctx.saved_tensors
        B, V, ldv, pad_idx = ctx.cfg
        one = torch.ones(1, device=lg.device)
        dl = torch.empty_like(lg)
        call('sf_speaker_glue_bwd', B, V, ldv, ptr(lg), ptr(target), pad_idx, ptr(one), ptr(dl),
             stream())
        return dl[:, :V] * dnll.reshape(B, 1), None, None, None, None, None, None


class _ScoredInstruction(LazyDict):
    """One output of Seq2SeqSpeaker._score_obs_actions_and_instructions (speaker.py:139-141, 198-202): 'instr_id' and
    'score' are there; 'word_indices', 'scores' and 'words' are cut from the batch's arrays when somebody reads them
    (the pragmatic re-ranking reads the score of 2 500 routes and nothing else, rational_follower.py:70-74)."""
    __slots__ = ('_w', '_s', '_m', '_tok')

    def __init__(self, instr_id, score, words_row, scores_row, m, tok):
        LazyDict.__init__(self, {'instr_id': instr_id, 'score': score}, ('word_indices', 'scores', 'words'))
        self._w, self._s, self._m, self._tok = words_row, scores_row, m, tok

    def _make(self, key):
        if key == 'scores':
            return self._s[:self._m].tolist()
        wi = self['word_indices'] if key == 'words' else self._w[:self._m].tolist()
        if key == 'words' and self._tok is not None:
            return self._tok.decode_sentence(wi, break_on_eos=True, join=False)
        return wi


class _PendingScores:
    """A chunked scoring sweep that has been issued and not collected yet (Seq2SeqSpeaker._issue_scores)."""
    obs_lists = None

    def matches(self, path_obs, feedback):
        return (self.obs_lists is not None and feedback == self.feedback and len(path_obs) == len(self.obs_lists)
                and all(a is b for a, b in zip(path_obs, self.obs_lists)))


def _ragged(counts):
    """[0..c0-1, 0..c1-1, ...] and the index of the owner of each element."""
    owner = np.repeat(np.arange(len(counts)), counts)
    starts = np.cumsum(counts) - counts
    return np.arange(len(owner)) - starts[owner], owner


class Seq2SeqSpeaker(object):
    """speaker.py:34-410 (teacher / argmax / sample decoding, scoring, train, save/load)."""
    feedback_options = ['teacher', 'argmax', 'sample']

    def __init__(self, env, results_path, encoder, decoder, instruction_len, max_episode_len=10):
        self.env = env
        self.results_path = results_path
        random.seed(1)
        self.results = {}
        self.losses = []
        self.encoder, self.decoder = encoder, decoder
        self.instruction_len = instruction_len
        self.max_episode_len = max_episode_len
        self.feedback = 'argmax'
        self._sample_seed = torch.initial_seed() & 0xFFFFFFFF     # `sample` feedback: counter-based draws (sf_sampling.h)
        self._sample_count = 0

    def write_results(self):
        with open(self.results_path, 'w') as f:
            json.dump(self.results, f)

    def _device(self):
        return next(self.decoder.parameters()).device

    def _batch_observations_and_actions(self, path_obs, path_actions, encoded_instructions):
        """speaker.py:68-121."""
        seq_lengths = np.array([len(a) for a in path_actions])
        Tp = int(seq_lengths.max())
        B = len(path_obs)
        dim = path_obs[0][0]['action_embedding'].shape[-1]
        fshape = path_obs[0][0]['feature'][0].shape
        mask = np.ones((B, Tp), np.uint8)
        acts = [np.zeros((B, dim), np.float32) for _ in range(Tp)]
        feats = [np.zeros((B,) + fshape, np.float32) for _ in range(Tp)]
        for i, (obs, actions) in enumerate(zip(path_obs, path_actions)):
            assert len(obs) == len(actions) + 1
            mask[i, :len(actions)] = 0
            for t, (ob, a) in enumerate(zip(obs[:-1], actions)):
                assert a >= 0
                feats[t][i] = ob['feature'][0]
                acts[t][i] = ob['action_embedding'][a]
        dev = self._device()
        to = lambda x: torch.from_numpy(x).to(dev)  # noqa: E731
        return ([obs[0] for obs in path_obs], [to(f) for f in feats], [to(a) for a in acts],
                to(mask), list(seq_lengths), encoded_instructions, list(range(B)))

    def _env_store(self):
        feats = getattr(self.env, 'image_features_list', None) or [None]
        return getattr(feats[0], 'store', None) or getattr(self, 'store', None)

    SCORE_CHUNK = 128          # rows of one persistent word-loop launch (csrc/sf_persist.hip)
    # large teacher-forced scoring sweeps: the full chunks as replayed graphs on two streams (speaker.SpeakerSweep) instead
    # of launch-by-launch issue: -3.5 ms per minibatch of 64 instructions once every (stream, path-step count) graph is
    # captured, ~0.3 s of captures before that (tools/pragmatic_scoring_ab.py): pays after ~85 minibatches.  Off by default.
    score_graphs = False

    @gc_paused
    def _score_on_device(self, path_obs, path_actions, encoded_instructions, feedback, store):
        """The same scoring over INDEX-FORM observations (no 'feature' / 'action_embedding' rows: an env that
        carries its feature store): the fused speaker engine (speaker.SpeakerEngine -- per path step visual
        attention + LSTMCell with in-kernel gathers, the word loop as one persistent launch in inference).  Results in
        the layout of the host path below.

        A large inference batch (the pragmatic re-ranking hands over ALL ~2 500 candidate routes of a minibatch at
        once, rational_follower.py:67-69) is walked in chunks of SCORE_CHUNK rows, each one persistent launch: the
        host packs chunk k + 1 while the device decodes chunk k, one download and one fault check at the end.  Rows
        are independent; the loss (a mean over ALL rows per step, up to the first step at which every row has ended,
        speaker.py:188-197) is re-assembled from the chunks' (sum, count) tables."""
        from . import speaker as spk
        import time
        marks = getattr(self, 'score_marks', None)       # tools/pragmatic_profile.py: wall-clock marks of the phases
        mark = (lambda name: marks.append((name, time.perf_counter()))) if marks is not None else (lambda name: None)
        mark('start')
        B = len(path_obs)
        # (a search's RouteObservations knows its instruction without building its dictionaries)
        instr_ids = [obs.instr_id if hasattr(obs, 'instr_id') else obs[0]['instr_id'] for obs in path_obs]
        pend = self.__dict__.pop('_pending_scores', None)
        if pend is not None:
            if pend.matches(path_obs, feedback) and not torch.is_grad_enabled():   # issued by the search itself: collect
                self.prefetch_hits = getattr(self, 'prefetch_hits', 0) + 1
                return self._finish_scores(pend, instr_ids, mark)
            torch.cuda.synchronize()                      # (not the routes it was issued for: let it drain, drop it)
        n = np.array([len(a) for a in path_actions], np.int32)
        # one row (vp_row, viewIndex, absViewIndex, rel_heading, rel_elevation, is_stop) per (observation, action) pair,
        # formed once per DISTINCT pair: the candidate routes of a search share their observation dictionaries
        seen = {}

        def pair_rows(lo, hi):                                        # [sum of n[lo:hi], 6], candidate-major
            rows = []
            for obs, actions in zip(path_obs[lo:hi], path_actions[lo:hi]):
                assert len(obs) == len(actions) + 1
                for t, a in enumerate(actions):
                    ob = obs[t]
                    k = (id(ob), a)
                    r = seen.get(k)
                    if r is None:
                        assert a >= 0
                        if a > 0:
                            d = ob['adj_loc_list'][a]
                            r = (ob['vp_row'], ob['viewIndex'], d['absViewIndex'], d['rel_heading'], d['rel_elevation'], 0.0)
                        else:
                            r = (ob['vp_row'], ob['viewIndex'], 0, 0.0, 0.0, 1.0)
                        seen[k] = r
                    rows.append(r)
            return np.array(rows, np.float64).reshape(-1, 6)
        return self._score_index_routes(n, pair_rows, encoded_instructions, instr_ids, feedback, store, mark)

    def _score_index_routes(self, n, pair_rows, encoded_instructions, instr_ids, feedback, store, mark=lambda name: None):
        """The scoring pass over routes in INDEX FORM: n [routes] steps each, pair_rows(lo, hi) = the [sum n[lo:hi], 6] rows
        (vp_row, viewIndex, absViewIndex, rel_heading, rel_elevation, is_stop) of routes lo..hi-1, route-major."""
        from . import speaker as spk
        B = len(n)
        if getattr(self, '_engine', None) is None or self._engine.store is not store:
            self._engine = spk.SpeakerEngine(self.encoder, self.decoder, store)
        eng = self._engine
        training = self.decoder.training
        chunked = B > 2 * self.SCORE_CHUNK and not training and not torch.is_grad_enabled() and eng.group is None
        if chunked:          # (pairs per chunk: formed while the device decodes the previous one)
            return self._finish_scores(self._issue_scores(n, pair_rows, encoded_instructions, feedback, store, mark),
                                       instr_ids, mark)
        S = self._score_steps(encoded_instructions, feedback)
        batch = spk.DeviceSpeakerBatch.from_synth(self._index_batch(n, pair_rows, encoded_instructions, 0, B),
                                                  device=store.device, max_length=self.instruction_len)
        st = eng.run(batch, S, feedback, train=training)                   # (fault check + per-step re-issue inside)
        both = torch.cat((st.words[1:].to(torch.float32), st.step_scores), dim=0).cpu().numpy()   # (the one host sync)
        mark('downloaded')
        return self._score_outputs(instr_ids, both, S, st.loss, None, store.device, mark)

    def _score_steps(self, encoded_instructions, feedback):
        S = self.instruction_len
        if feedback == 'teacher':
            # the reference's loop ends once every row has ended (speaker.py:199-200): with teacher forcing that is the
            # longest instruction's EOS -- steps behind it add nothing to any score or to the loss
            S = max(1, min(S, max(len(e_) for e_ in encoded_instructions) + 1))
        return S

    @staticmethod
    def _index_batch(n, rows_of, encoded_instructions, lo, hi):
        """synth.SpeakerBatch (index form, speaker.py:68-121) of routes lo..hi-1: rows_of(lo, hi) = their [sum n, 6]
        (vp_row, viewIndex, absViewIndex, rel_heading, rel_elevation, is_stop) rows, route-major."""
        from . import synth
        m = n[lo:hi]
        Tp, Bc = int(m.max()), hi - lo
        t_of, owner = _ragged(m)
        at = t_of * Bc + owner
        part = rows_of(lo, hi)

        def grid(col, dt, fill=0):
            out = np.full(Tp * Bc, fill, dt)
            out[at] = part[:, col]
            return out.reshape(Tp, Bc)
        return synth.SpeakerBatch(instr=list(encoded_instructions[lo:hi]), path_len=m, vp=grid(0, np.int32),
                                  view=grid(1, np.int32), act_view=grid(2, np.int32), act_heading=grid(3, np.float64),
                                  act_elevation=grid(4, np.float64), act_is_stop=grid(5, np.float64, 1.0) != 0)

    def _issue_scores(self, n, rows_of, encoded_instructions, feedback, store, mark=lambda name: None):
        """Issues the chunked scoring sweep (no host sync) and returns what _finish_scores needs to collect it."""
        from . import speaker as spk
        if getattr(self, '_engine', None) is None or self._engine.store is not store:
            self._engine = spk.SpeakerEngine(self.encoder, self.decoder, store)
        eng = self._engine
        B = len(n)
        S = self._score_steps(encoded_instructions, feedback)

        def staged_batch(k, lo, hi):
            """Chunk k through ONE pinned staging slice and ONE asynchronous H2D copy (a pageable `.to(device)` is a
            blocking copy: it would wait for the previous chunk's launches and serialise host and device)."""
            sb = self._index_batch(n, rows_of, encoded_instructions, lo, hi)
            Tp, Bc = sb.vp.shape
            nbytes = spk.packed_layout(Bc, Tp, self.instruction_len)['bytes']
            pin, dev_buf = self._staging(store.device, k, nbytes)
            spk.pack_speaker_batch(sb, pin.numpy(), Lmax=self.instruction_len, Tp=Tp)
            dev_buf[:nbytes].copy_(pin[:nbytes], non_blocking=True)
            return spk.batch_from_packed(dev_buf, Bc, Tp, Lmax=self.instruction_len, row0=lo)

        def sweep():
            parts = []
            for k, lo in enumerate(range(0, B, self.SCORE_CHUNK)):
                st = eng.score(staged_batch(k, lo, min(lo + self.SCORE_CHUNK, B)), S, feedback, train=False)
                parts.append((st.words[1:].to(torch.float32), st.step_scores, st.sum_cnt))
            return parts
        pend = _PendingScores()
        pend.B, pend.S, pend.feedback, pend.store, pend.sweep, pend.site = B, S, feedback, store, sweep, eng.site_next
        pend.graphs = None
        n_full = B // self.SCORE_CHUNK
        if self.score_graphs and feedback == 'teacher' and n_full >= 4 and eng.persistent and eng.group is None:
            # the full chunks as replayed graphs on two streams (speaker.SpeakerSweep: one captured pass per stream and
            # path-step count; the encoder steps of one chunk run beside the other stream's word recurrence), the
            # remainder issued launch by launch behind them
            key = (id(store), S, self.instruction_len)
            sw = self.__dict__.setdefault('_score_sweeps', {}).get(key)
            if sw is None or sw.enc is not self.encoder or sw.dec is not self.decoder:
                sw = self._score_sweeps[key] = spk.SpeakerSweep(self.encoder, self.decoder, store, self.SCORE_CHUNK, S,
                                                                feedback='teacher', Lmax=self.instruction_len, with_scores=True)
            full = [self._index_batch(n, rows_of, encoded_instructions, lo, lo + self.SCORE_CHUNK)
                    for lo in range(0, n_full * self.SCORE_CHUNK, self.SCORE_CHUNK)]
            pend.graphs = (sw, sw.issue(full))
            lo = n_full * self.SCORE_CHUNK
            pend.parts = []
            if lo < B:
                st = eng.score(staged_batch(0, lo, B), S, feedback, train=False)
                pend.parts.append((st.words[1:].to(torch.float32), st.step_scores, st.sum_cnt))
        else:
            pend.parts = sweep()
        mark('issued')
        return pend

    def _finish_scores(self, pend, instr_ids, mark=lambda name: None):
        from .runtime import take_fault
        eng, dev = self._engine, pend.store.device
        swept = None
        if pend.graphs is not None:
            sw, handle = pend.graphs
            swept = sw.finish(handle, check_faults=False)                     # (waits for its two streams)
        bits = take_fault(dev)                                                # (one sync for the whole sweep)
        mark('device done')
        parts = pend.parts
        if bits:                                                              # a starved persistent launch: per-step kernels
            swept = None                                                      # (everything again, launch by launch)
            eng.fallbacks += 1
            keep, eng.persistent, eng.site_next = eng.persistent, False, pend.site
            try:
                parts = pend.sweep()
            finally:
                eng.persistent = keep
            again = take_fault(dev)
            if again:
                raise PersistentLaunchFault('fault bits %d, and %d after the per-step re-issue' % (bits, again))
        if swept is not None:
            w16, sc, cnt = swept                                              # [n,S,128] int16 / f32, [n,S,2]
            S_ = pend.S
            words_h = w16.transpose(1, 0, 2).reshape(S_, -1).astype(np.float32)
            scores_h = sc.transpose(1, 0, 2).reshape(S_, -1)
            sum_cnt = cnt.astype(np.float32).sum(0)
            if parts:
                tail = torch.cat((parts[0][0], parts[0][1]), dim=0).cpu().numpy()
                words_h = np.concatenate((words_h, tail[:S_]), axis=1)
                scores_h = np.concatenate((scores_h, tail[S_:]), axis=1)
                # (float32 sums chunk by chunk, as the stacked device sum does)
                sum_cnt = sum_cnt + parts[0][2].cpu().numpy()
            both = np.concatenate((words_h, scores_h), axis=0)
        else:
            both = torch.cat((torch.cat([p_[0] for p_ in parts], dim=1), torch.cat([p_[1] for p_ in parts], dim=1)),
                             dim=0).cpu().numpy()
            sum_cnt = torch.stack([p_[2] for p_ in parts]).sum(0).cpu().numpy()        # [S,2]: all rows
        mark('downloaded')
        return self._score_outputs(instr_ids, both, pend.S, None, sum_cnt, dev, mark)

    def _score_outputs(self, instr_ids, both, S, loss, sum_cnt, device, mark=lambda name: None):
        B = len(instr_ids)
        words, sc = both[:S].T.astype(np.int64), np.ascontiguousarray(both[S:].T)                  # [B,S] each
        is_eos = words == EOS
        m_all = np.where(is_eos.any(1), is_eos.argmax(1) + 1, S)           # up to and including the first EOS
        if loss is None:
            # speaker.py:188-197 over the whole batch: step means up to and including the first step at which every row
            # has produced EOS, summed in step order in float32 (what sf_speaker_loss_finalize does for one batch)
            last = int(m_all.max()) - 1
            total = np.float32(0)
            for t in range(last + 1):
                if sum_cnt[t, 1] > 0:
                    total = np.float32(total + np.float32(sum_cnt[t, 0] / sum_cnt[t, 1]))
            loss = torch.tensor(float(total), device=device)
        totals = np.cumsum(sc, axis=1, dtype=np.float32)                   # (sequential float32 partial sums)
        tok = getattr(self.env, 'tokenizer', None)
        ends = m_all.tolist()
        last = totals[np.arange(B), m_all - 1].tolist()
        outputs = [_ScoredInstruction(instr_ids[i], last[i], words[i], sc[i], ends[i], tok) for i in range(B)]
        mark('outputs')
        return outputs, loss

    def route_scores_hook(self, feedback='teacher'):
        """For Seq2SeqAgent.candidates_hook (frontier.state_factored_search): the search hands over its completed routes
        in index form -- n [routes] steps each, rows [sum n, 6], the instruction of every route -- the moment its last
        iteration is done, BEFORE it builds their result dictionaries; the scoring sweep is issued right there, so the
        device decodes while the host builds ~2 500 dictionaries.  The hook returns `bind`: called with the routes'
        observation lists once they exist; the next _score_obs_actions_and_instructions over exactly those lists
        collects the scores instead of issuing them again (rational_follower.py:67-69 is that call)."""
        def hook(n, rows, instructions):
            store = self._env_store()
            if store is None or len(n) <= 2 * self.SCORE_CHUNK or self.decoder.training:
                return None
            stale = self.__dict__.pop('_pending_scores', None)
            if stale is not None:
                torch.cuda.synchronize()
            from . import speaker as spk
            if getattr(self, '_engine', None) is None or self._engine.store is not store:
                self._engine = spk.SpeakerEngine(self.encoder, self.decoder, store)
            if self._engine.group is not None:
                return None
            first = np.concatenate(([0], np.cumsum(n)))
            with torch.no_grad():       # (an inference sweep; a caller that wants gradients does not collect it)
                pend = self._issue_scores(np.asarray(n, np.int32), lambda lo, hi: rows[first[lo]:first[hi]], instructions,
                                          feedback, store)

            def bind(obs_lists):
                pend.obs_lists = list(obs_lists)
                self._pending_scores = pend
            return bind
        return hook

    def _staging(self, device, k, nbytes):
        """Pinned host / device staging buffers of chunk k of a chunked scoring pass (kept across calls; one pair per
        chunk, so that a chunk's buffer is never rewritten while its copy or its kernels are in flight)."""
        pool = self.__dict__.setdefault('_stage_pool', [])
        while len(pool) <= k:
            pool.append(None)
        if pool[k] is None or pool[k][0].numel() < nbytes or pool[k][1].device != device:
            cap = max(nbytes, 1 << 17)
            pool[k] = (torch.empty(cap, dtype=torch.uint8).pin_memory(), torch.empty(cap, dtype=torch.uint8, device=device))
        return pool[k]

    def _score_obs_actions_and_instructions(self, path_obs, path_actions, encoded_instructions,
                                            feedback):
        """speaker.py:123-202."""
        assert len(path_obs) == len(path_actions) == len(encoded_instructions)
        if 'feature' not in path_obs[0][0]:
            store = self._env_store()
            if store is None:
                raise RuntimeError('index-form observations need a features.FeatureStore (env.image_features_list[0].store '
                                   'or speaker.store)')
            return self._score_on_device(path_obs, path_actions, encoded_instructions, feedback, store)
        start_obs, feats, acts, path_mask, _, encoded_instructions, perm = \
            self._batch_observations_and_actions(path_obs, path_actions, encoded_instructions)
        dev = self._device()
        instr_seq, _, _ = batch_instructions_from_encoded(encoded_instructions, self.instruction_len,
                                                         device=dev)
        B = len(start_obs)
        ctx, h, c = self.encoder(acts, feats)
        w_t = torch.full((B,), BOS, dtype=torch.int64, device=dev)
        ended = np.array([False] * B)
        ended_dev = torch.zeros(B, dtype=torch.uint8, device=dev)
        outputs = [{'instr_id': start_obs[i]['instr_id'], 'word_indices': [], 'scores': []}
                   for i in range(B)]
        loss = 0
        seq_scores = torch.zeros(B, device=dev)
        for t in range(self.instruction_len):
            h, c, alpha, logit = self.decoder(w_t.view(-1, 1), h, c, ctx, path_mask)
            target = instr_seq[:, t].contiguous()
            self._sample_count += 1
            nll, live, w_t, score = _SpeakerGlueFn.apply(logit, target, FEEDBACK[feedback],
                                                         ended_dev, PAD, EOS, (self._sample_seed, self._sample_count))
            seq_scores = seq_scores + score
            loss = loss + nll.sum() / live.sum().clamp(min=1.0)
            words, sc, ss = w_t.tolist(), score.tolist(), seq_scores.tolist()
            for i in range(B):
                if not ended[i]:
                    outputs[i]['word_indices'].append(int(words[i]))
                    outputs[i]['score'] = float(ss[i])
                    outputs[i]['scores'].append(sc[i])
                if words[i] == EOS:
                    ended[i] = True
            if ended.all():
                break
        tok = getattr(self.env, 'tokenizer', None)
        for item in outputs:
            item['words'] = (tok.decode_sentence(item['word_indices'], break_on_eos=True, join=False)
                             if tok is not None else list(item['word_indices']))
        return outputs, loss

    index_gold_routes = True   # rollout(): the minibatch's gold routes from the navigation tables (nav.NavTable.gold_routes)
                               # instead of a lock-step walk of the host environment

    def rollout(self, load_next_minibatch=True):
        store = self._env_store()
        if (self.index_gold_routes and store is not None and getattr(self.env, 'host_table', 1) is None
                and hasattr(self.env, 'graphs')):
            # speaker.py:348-360 without the per-step host loop of env.py:823-848: teacher actions and transitions are
            # table look-ups over the whole minibatch, the routes arrive in index form (what the device consumes)
            from .nav import table_for
            self.env.reset(load_next_minibatch=load_next_minibatch)
            items = list(self.env.batch)
            n, rows = table_for(self.env, store).gold_routes(items, self.max_episode_len)
            first = np.concatenate(([0], np.cumsum(n)))
            outputs, loss = gc_paused(self._score_index_routes)(
                n, lambda lo, hi: rows[first[lo]:first[hi]], [it['instr_encoding'] for it in items],
                [it['instr_id'] for it in items], self.feedback, store)
            self.loss = loss
            self.losses.append(float(loss.detach()) if torch.is_tensor(loss) else float(loss))
            return outputs
        path_obs, path_actions, enc = self.env.gold_obs_actions_and_instructions(
            self.max_episode_len, load_next_minibatch=load_next_minibatch)
        outputs, loss = self._score_obs_actions_and_instructions(path_obs, path_actions, enc,
                                                                 self.feedback)
        self.loss = loss
        self.losses.append(float(loss.detach()) if torch.is_tensor(loss) else float(loss))
        return outputs

    def beam_search(self, beam_size, path_obs, path_actions):
        """speaker.py:211-318 (search.py)."""
        from . import search
        return search.speaker_beam_search(self, beam_size, path_obs, path_actions)

    sweep_test_after = 30            # test(): minibatches (cumulative over calls) before the split is decoded as a sweep

    def _test_as_a_sweep(self):
        """test() over an index-form environment (speaker.py:397-414; data_augmentation_from_speaker.py's loop) through
        speaker.SpeakerSweep: the epoch's minibatches are drawn first (the same env.reset calls, the same use of
        `random` where the epoch wraps), their gold routes come from the navigation tables, and the device decodes them
        as replayed graphs on two streams -- packing, copies and decoding overlapped -- instead of one launch-by-launch
        pass per minibatch (4.6 -> 2.1 ms per minibatch of 100).  The graphs of a sweep (one per stream and route
        length) cost ~40 ms once: taken once this agent's test() calls have seen `sweep_test_after` minibatches (a
        data-augmentation pass at once, periodic validations from their second).  False: not applicable, the caller
        loops over rollout()."""
        store = self._env_store()
        if not (self.beam_size == 1 and self.index_gold_routes and store is not None and not self.decoder.training
                and self.feedback == 'argmax' and getattr(self.env, 'host_table', 1) is None
                and hasattr(self.env, 'graphs') and hasattr(self.env, 'data') and store.device.type == 'cuda'):
            return False
        from . import speaker as spk
        B = self.env.batch_size
        n_batches = -(-len(self.env.data) // B)
        seen = self.__dict__.get('_test_minibatches', 0)
        self._test_minibatches = seen + n_batches
        if seen + n_batches < self.sweep_test_after:
            return False
        drawn, ids, looped = [], set(), False
        while not looped:                                      # (the loop of speaker.py:404-413 over the items themselves)
            self.env.reset()
            items = list(self.env.batch)
            drawn.append(items)
            for it in items:
                if it['instr_id'] in ids:
                    looped = True
                ids.add(it['instr_id'])
        key = (id(store), B, self.instruction_len)
        cached = self.__dict__.get('_test_sweep')
        if cached is None or cached[0] != key:
            cached = self._test_sweep = (key, spk.SpeakerSweep(self.encoder, self.decoder, store, B, self.instruction_len,
                                                               feedback='argmax', Lmax=self.instruction_len,
                                                               with_scores=True), store)
        sweep = cached[1]
        words, scores, cnt = sweep.run([self._routes_of(items, store)[0] for items in drawn])
        S = self.instruction_len
        for k, items in enumerate(drawn):
            both = np.concatenate((words[k].astype(np.float32), scores[k]), axis=0)
            outputs, loss = self._score_outputs([it['instr_id'] for it in items], both, S, None, cnt[k], store.device)
            self.loss = loss
            self.losses.append(float(loss))
            for result in outputs:
                if result['instr_id'] not in self.results:
                    self.results[result['instr_id']] = result
        return True

    def test(self, use_dropout=False, feedback='argmax', allow_cheat=False, beam_size=1):
        if not allow_cheat:
            assert feedback in ['argmax', 'sample']
        self.feedback = feedback
        for m in (self.encoder, self.decoder):
            m.train() if use_dropout else m.eval()
        self.beam_size = beam_size
        self.env.reset_epoch()
        self.losses = []
        self.results = {}
        if self._test_as_a_sweep():
            return self.results
        with torch.no_grad():
            return self._test_loop()

    def _test_loop(self):
        looped = False
        while True:
            for result in self.rollout():
                if result['instr_id'] in self.results:
                    looped = True
                else:
                    self.results[result['instr_id']] = result
            if looped:
                break
        return self.results

    train_without_outputs = True     # train(): forward and backward issued back to back, ONE host sync per iteration

    def _routes_of(self, items, store):
        """Host side of a training minibatch: its gold routes from the navigation tables, in the index form the device
        consumes (synth.SpeakerBatch), and the number of word steps."""
        from .nav import table_for
        n, rows = table_for(self.env, store).gold_routes(items, self.max_episode_len)
        first = np.concatenate(([0], np.cumsum(n)))
        enc = [it['instr_encoding'] for it in items]
        return (self._index_batch(n, lambda lo, hi: rows[first[lo]:first[hi]], enc, 0, len(items)),
                self._score_steps(enc, self.feedback))

    def _train_iteration_on_index_routes(self):
        """speaker.py:376-395's rollout + backward for train(), which reads nothing but the loss: the scoring pass and its
        backward are issued back to back (no download of words and scores, no result dictionaries, no sync between
        them), the NEXT minibatch's routes are formed while the device works (peeked, not drawn), and the fault word of
        the persistent launches is read once, where the loss is read; a fault re-issues the SAME minibatch on the
        per-step kernels.  Same kernels, same dropout sites as `rollout()` + `loss.backward()`.  False: not applicable
        (dense environment, row-sharded engine): the caller takes that path."""
        store = self._env_store()
        if not (self.train_without_outputs and self.index_gold_routes and store is not None
                and getattr(self.env, 'host_table', 1) is None and hasattr(self.env, 'graphs')):
            return False
        from . import speaker as spk
        from .runtime import take_fault
        if getattr(self, '_engine', None) is None or self._engine.store is not store:
            self._engine = spk.SpeakerEngine(self.encoder, self.decoder, store)
        eng, dev = self._engine, store.device
        if eng.group is not None:
            return False
        self.env.reset()
        items = list(self.env.batch)
        ahead, prep = self.__dict__.pop('_routes_ahead', None), None
        if (ahead is not None and ahead[0] == self.feedback and len(ahead[1]) == len(items)
                and all(a is b for a, b in zip(ahead[1], items))):
            prep = ahead[2]
        sb, S = prep if prep is not None else self._routes_of(items, store)
        batch = spk.DeviceSpeakerBatch.from_synth(sb, device=dev, max_length=self.instruction_len)
        take_fault(dev)                                       # (whatever an earlier pass left behind is not ours)
        site = eng.site_next
        st = eng.score(batch, S, self.feedback, train=True)
        st.loss.backward()
        peek = getattr(self.env, 'peek_next_minibatch', None)
        nxt = peek(False) if peek is not None else None
        if nxt is not None:
            self._routes_ahead = (self.feedback, nxt, self._routes_of(nxt, store))
        if take_fault(dev):                                   # the iteration's host sync
            for m in (self.encoder, self.decoder):
                for p_ in m.parameters():
                    if p_.grad is not None:
                        p_.grad.zero_()
            keep, eng.persistent, eng.site_next = eng.persistent, False, site
            eng.fallbacks += 1
            try:
                st = eng.score(batch, S, self.feedback, train=True)
                st.loss.backward()
            finally:
                eng.persistent = keep
            if take_fault(dev):
                raise PersistentLaunchFault('the per-step re-issue of a training iteration raised a fault again')
        self.loss = st.loss
        self.losses.append(float(st.loss.detach()))
        return True

    def train(self, encoder_optimizer, decoder_optimizer, n_iters, feedback='teacher'):
        assert feedback in self.feedback_options
        self.feedback = feedback
        self.encoder.train()
        self.decoder.train()
        self.losses = []
        for _ in range(1, n_iters + 1):
            encoder_optimizer.zero_grad()
            decoder_optimizer.zero_grad()
            if self._train_iteration_on_index_routes():
                encoder_optimizer.step()
                decoder_optimizer.step()
                continue
            self.rollout()
            self.loss.backward()
            # teacher-forced passes run their recurrence (forward and backward) as persistent launches since round 5: a
            # starved BACKWARD has poisoned the gradients -- never step on them; train this iteration on the next
            # minibatch with the per-step kernels instead (the forward's own check sits inside SpeakerEngine.run)
            dev = next(self.decoder.parameters()).device
            if _persistent_fault(dev):
                encoder_optimizer.zero_grad()
                decoder_optimizer.zero_grad()
                self.losses.pop()
                eng = getattr(self, '_engine', None)
                keep = eng.persistent if eng is not None else True
                if eng is not None:
                    eng.persistent = False
                    eng.fallbacks += 1
                try:
                    self.rollout()
                    self.loss.backward()
                finally:
                    if eng is not None:
                        eng.persistent = keep
                if _persistent_fault(dev):
                    raise PersistentLaunchFault('the per-step re-issue of a training iteration raised a fault again')
            encoder_optimizer.step()
            decoder_optimizer.step()

    def _encoder_and_decoder_paths(self, base_path):
        return base_path + '_enc', base_path + '_dec'

    def save(self, path):
        ep, dp = self._encoder_and_decoder_paths(path)
        torch.save(self.encoder.state_dict(), ep)
        torch.save(self.decoder.state_dict(), dp)

    def load(self, path, **kwargs):
        ep, dp = self._encoder_and_decoder_paths(path)
        self.encoder.load_state_dict(torch.load(ep, **kwargs))
        self.decoder.load_state_dict(torch.load(dp, **kwargs))
