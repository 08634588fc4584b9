"""Probe: why is the eager speaker training iteration host-bound (13 ms) behind another measurement in the same process
when it takes 1.8 ms of host time in a fresh one?"""
import os, sys, torch, time, cProfile, pstats
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))); sys.argv = ['bench.py']
import bench
from speaker_follower_amd import synth, features, follower, bench_extras, speaker, optim, dp
dev = torch.device('cuda', 0)
store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
def spk(tag, prof=False):
    senc, sdec = bench_extras._speaker_models(dev); senc.train(); sdec.train()
    sb = synth.speaker_batch(seed=0, batch=100, n_viewpoints=10567, min_path=4, max_path=7, min_len=10, max_len=79)
    b = speaker.DeviceSpeakerBatch.from_synth(sb, device=dev)
    pe = [p for p in senc.parameters() if p.requires_grad]; pd = [p for p in sdec.parameters() if p.requires_grad]
    flat = dp.FlatGrads(pe + pd)
    oe, od = optim.FusedAdam(pe, lr=1e-4, weight_decay=5e-4), optim.FusedAdam(pd, lr=1e-4, weight_decay=5e-4)
    eng = speaker.SpeakerEngine(senc, sdec, store)
    def it():
        flat.zero(); st = eng.score(b, 80, 'teacher', train=True); st.loss.backward(); oe.step(); od.step()
    for _ in range(3): it()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): it()
    host = (time.perf_counter() - t0) / 10; torch.cuda.synchronize(); wall = (time.perf_counter() - t0) / 10
    print(tag, 'speaker eager %.2f ms wall, %.2f host; reserved %.1f GB' % (1e3 * wall, 1e3 * host, torch.cuda.memory_reserved() / 2**30), flush=True)
    if prof:
        pr = cProfile.Profile(); pr.enable()
        for _ in range(5): it()
        torch.cuda.synchronize(); pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(10)
spk('fresh')
fb = synth.follower_batch(seed=0, batch=100, steps=20, n_viewpoints=10567)
batch = follower.DeviceFollowerBatch.from_synth(fb, device=dev)
enc, dec, _, _ = bench.build_models(101, dev)
t = bench.measure_train(enc, dec, store, batch, 20, 10, 5)
print('follower eager %.3f graph %.3f' % (t['eager']['ms_per_iteration'], t['ms_per_iteration']), flush=True)
spk('behind the follower measurement (graph captured)', prof=True)
