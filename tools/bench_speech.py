"""C4 probe: sonar_speech_encoder_eng (fbank + 24-layer conformer + 3-layer attention pooler),
64 clips x 10 s @ 16 kHz, fp16, 1 GPU."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd.speech_encoder import get_speech_encoder_config, SpeechEncoderEngine, waveforms_to_fbank_batch

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    cfg = get_speech_encoder_config("english")
    d, f = 1024, 4096
    dev = "cuda:0"
    from tools.synth import speech_encoder_state_dict
    sd = speech_encoder_state_dict(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    eng = SpeechEncoderEngine(cfg, sd, device=dev)
    del sd
    wavs = torch.rand(n, 160000, device=dev, generator=g) * 2 - 1
    def run():
        feats, _ = waveforms_to_fbank_batch(list(wavs))                         # [n, 998, 80], one launch
        return eng.forward(feats, None, torch.float16), feats
    run(); torch.cuda.synchronize()
    t0 = time.time(); reps = 3
    for _ in range(reps):
        emb, feats = run()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    # encoder only
    t1 = time.time()
    for _ in range(reps):
        eng.forward(feats, None, torch.float16)
    torch.cuda.synchronize()
    de = (time.time() - t1) / reps
    frames = n * 499
    print(f"speech n={n} x 10 s: total {dt*1e3:.1f} ms ({n/dt:.1f} clips/s, {n*10/dt:.0f}x real time); encoder only {de*1e3:.1f} ms; "
          f"{frames} stacked frames; emb finite={bool(torch.isfinite(emb).all())} checksum {float(emb.double().sum()):.6f}")

if __name__ == "__main__":
    main()
