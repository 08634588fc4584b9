"""Wall time of the speaker's training iteration (speaker.py:376-395: teacher-forced scoring of a minibatch of
100 paths x 80 words with dropout, backward, two Adam steps) through SpeakerEngine, and how much of it is host time."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch                # noqa: E402

import bench                # noqa: E402
from speaker_follower_amd import bench_extras, features, synth, speaker, optim, dp    # noqa: E402


def main():
    dev = torch.device('cuda', 0)
    store = features.FeatureStore(bench.device_table(10567, 1234, dev), device=dev)
    senc, sdec = bench_extras._speaker_models(dev)
    senc.train()
    sdec.train()
    sb = synth.speaker_batch(seed=0, batch=100, n_viewpoints=10567, min_path=4, max_path=7, min_len=10, max_len=79)
    batch = speaker.DeviceSpeakerBatch.from_synth(sb, device=dev)
    pe = [p for p in senc.parameters() if p.requires_grad]
    pd = [p for p in sdec.parameters() if p.requires_grad]
    flat = dp.FlatGrads(pe + pd)
    oe, od = optim.FusedAdam(pe, lr=1e-4, weight_decay=5e-4), optim.FusedAdam(pd, lr=1e-4, weight_decay=5e-4)
    eng = speaker.SpeakerEngine(senc, sdec, store)

    def it():
        flat.zero()
        st = eng.score(batch, 80, 'teacher', train=True)
        st.loss.backward()
        oe.step()
        od.step()
        return st
    for _ in range(3):
        it()
    torch.cuda.synchronize()
    n = 10
    t0 = time.perf_counter()
    for _ in range(n):
        st = it()
    host = (time.perf_counter() - t0) / n
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) / n
    if '--cprofile' in sys.argv:
        import cProfile
        import pstats
        pr = cProfile.Profile()
        pr.enable()
        for _ in range(5):
            it()
        torch.cuda.synchronize()
        pr.disable()
        pstats.Stats(pr).sort_stats('tottime').print_stats(22)
    print('speaker training iteration (B = 100, 80 words, teacher forcing, dropout 0.5): %.2f ms wall, %.2f ms of host issue, '
          'loss %.4f' % (1e3 * wall, 1e3 * host, float(st.loss.detach())))


if __name__ == '__main__':
    main()
