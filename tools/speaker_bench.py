"""Speaker scoring throughput: B=100 paths of 4-7 steps, 80 word steps, teacher-forced
NLL (what the pragmatic re-ranking runs per candidate) and greedy decoding."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from speaker_follower_amd import synth, model, features, speaker
d = synth.FULL
enc_w, dec_w = synth.speaker_weights(5)
enc = model.SpeakerEncoderLSTM(d.feat, d.feat, d.hidden, 0.5)
dec = model.SpeakerDecoderLSTM(d.vocab, d.word, d.hidden, 0.5, glove=dec_w['embedding.weight'])
enc.load_state_dict({k: torch.tensor(v) for k, v in enc_w.items()}); dec.load_state_dict({k: torch.tensor(v) for k, v in dec_w.items()})
enc.cuda().eval(); dec.cuda().eval()
B, S, NVP = 100, 80, 2048
sb = synth.speaker_batch(seed=0, batch=B, n_viewpoints=NVP, min_path=4, max_path=7, min_len=10, max_len=79)
store = features.FeatureStore(synth.feature_table(0, NVP))
batch = speaker.DeviceSpeakerBatch.from_synth(sb)
eng = speaker.SpeakerEngine(enc, dec, store)
for fb in ('teacher', 'argmax'):
    replay, gst = eng.capture(batch, S, fb)
    for _ in range(3): replay()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): replay()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print('%-8s %.3f ms per batch of %d x %d word steps  -> %.0f word-steps/s (hipGraph replay)' % (fb, dt * 1e3, B, S, B * S / dt))
    with torch.no_grad():
        for _ in range(3): st = eng.score(batch, S, fb, train=False)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10): st = eng.score(batch, S, fb, train=False)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print('%-8s %.3f ms per batch of %d x %d word steps  -> %.0f word-steps/s (eager issue)' % (fb, dt * 1e3, B, S, B * S / dt))
