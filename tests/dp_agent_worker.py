"""Worker of tests/test_gpu_dataparallel.py::test_agent_train_under_a_process_group_replays_segments (its own process: it
initialises a process group).  Seq2SeqAgent.train on the device environment with optim.FusedAdam, once without a group (every
iteration ONE graph replay) and once with a one-rank gloo group whose collectives are all issued (dp.FORCE_COLLECTIVES) and
gradient buckets on the engine: the iterations replay as SEGMENTS cut at the collective points.  A one-rank all-reduce is the
identity, so losses and weights must agree.  Prints one JSON line."""
import json
import os
import random
import socket
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))


def main():
    import torch
    import torch.distributed as dist
    import search_world as W
    from speaker_follower_amd import agents, model, optim, synth, features, nav, dp
    torch.cuda.set_device(0)
    dev = torch.device('cuda', 0)
    env, table = W.build_world(dense=False, n_items=36, batch=12, item_seed=77)
    store = features.FeatureStore(table, device=dev)
    nt = nav.NavTable(env, store)
    d = synth.FULL
    enc_w, dec_w = synth.follower_weights_peaky(303)
    order = list(env.data)

    def agent(group):
        enc = model.EncoderLSTM(d.vocab, d.word, d.hidden, 0, 0.5, glove=enc_w['embedding.weight'])
        dec = model.AttnDecoderLSTM(d.feat, d.hidden, 0.5, feature_size=d.feat)
        enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()})
        dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
        enc.to(dev)
        dec.to(dev)
        ag = agents.Seq2SeqAgent(env, '/tmp/sf_dp_agent.json', enc, dec, episode_len=6)
        ag.store = store
        ag.use_device_env(nt)
        if group is not None:
            ag._engine.group = group
            ag._engine.grad_sync = dp.BucketedGrads(dp.follower_buckets(enc, dec), group=group)
        oe = optim.FusedAdam([p for p in enc.parameters() if p.requires_grad], lr=1e-4, weight_decay=5e-4)
        od = optim.FusedAdam([p for p in dec.parameters() if p.requires_grad], lr=1e-4, weight_decay=5e-4)
        env.data[:] = order
        random.seed(11)
        env.reset_epoch()
        torch.manual_seed(4)
        ag._engine.dropout_seed = 4242
        ag.train(oe, od, 5, feedback='teacher')
        torch.cuda.synchronize()
        w = torch.cat([p.detach().reshape(-1) for m in (enc, dec) for p in m.parameters()]).clone()
        tg = ag._train_graph_state[1]
        return list(ag.losses), w, (len(tg.segments) if tg.segments is not None else 0), tg.replays, ag._engine.fallbacks

    plain = agent(None)
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK='0', WORLD_SIZE='1')
    dist.init_process_group('gloo')
    dp.FORCE_COLLECTIVES = True
    grouped = agent(dist.group.WORLD)
    scale = float(plain[1].abs().max())
    out = dict(losses_plain=plain[0], losses_group=grouped[0], segments=[plain[2], grouped[2]], replays=[plain[3], grouped[3]],
               fallbacks=[plain[4], grouped[4]], weight_rel_diff=float((plain[1] - grouped[1]).abs().max()) / scale,
               finite=bool(torch.isfinite(grouped[1]).all()))
    dist.destroy_process_group()
    print('DP_AGENT_WORKER ' + json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
