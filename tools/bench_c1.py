"""C1 probe (BASELINE configs[0]): text_sonar_basic_encoder fp16, 32 sentences, lengths randint(8,65) seed 0, on one GPU.
Prints ms per forward and the per-kernel HIP-event profile of the engine.  usage: python tools/bench_c1.py [n_sentences] [reps]"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd.text_encoder import SonarTextTransformerEncoderModel, SequenceBatch, PaddingMask, get_text_encoder_config
from tools.synth import text_encoder_state_dict, V


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    dev = "cuda:0"
    model = SonarTextTransformerEncoderModel(get_text_encoder_config("basic"), text_encoder_state_dict(dev), device=dev,
                                             dtype=torch.float16, fp16_residual=True)
    g = torch.Generator().manual_seed(0)
    lens = torch.randint(8, 65, (n,), generator=g).to(torch.int32)
    ids = torch.randint(4, 256001, (n, int(lens.max())), generator=g)
    batch = SequenceBatch(ids.to(dev), PaddingMask(lens, ids.shape[1]))
    for _ in range(3):
        model(batch)
    torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(reps):
        out = model(batch).sentence_embeddings
    torch.cuda.synchronize()
    dt = (time.time() - t0) / reps
    model.engine.set_profiling(True)
    for _ in range(5):
        model(batch)
    torch.cuda.synchronize()
    model.engine.set_profiling(False)
    prof = model.engine.read_profile()
    per = {k: round(v["ms"] / 5 * 1e3, 1) for k, v in prof.items()}
    print(f"c1 n={n} tokens={int(lens.sum())}: {dt * 1e3:.3f} ms per forward ({n / dt:.0f} sentences/s); us per forward by kernel: {per}; "
          f"sum {sum(per.values()):.0f} us; finite={bool(torch.isfinite(out).all())}")


if __name__ == "__main__":
    main()
