"""C5 probe over decode-chain configurations in ONE process (one engine): each configuration is a chain count plus
`SMI_DEC_*` switches (read by the engine at the start of every generate call).
usage: python tools/bench_decoder_chains.py [n steps] -- "chains=2" "chains=2 SMI_DEC_KS_FFN=16" ...
Prints ms per step per configuration and whether the best hypotheses equal those of the first configuration."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd.text_decoder import TextDecoderEngine, get_text_decoder_config  # noqa: E402

KNOBS = ("SMI_DEC_KS_OUT", "SMI_DEC_KS_FFN", "SMI_DEC_FFN1_ENGINE", "SMI_DEC_LOGITS_GRID", "SMI_G2_SPLITK_MIN")


def main():
    argv = sys.argv[1:]
    cfgs = ["chains=1", "chains=2", "chains=3"]
    if "--" in argv:
        cfgs = argv[argv.index("--") + 1:]
        argv = argv[:argv.index("--")]
    n = int(argv[0]) if len(argv) > 0 else 256
    steps = int(argv[1]) if len(argv) > 1 else 64
    reps = int(argv[2]) if len(argv) > 2 else 3
    dev = "cuda:0"
    from tools.synth import text_decoder_state_dict
    eng = TextDecoderEngine(get_text_decoder_config("basic"), text_decoder_state_dict(dev), device=dev)
    torch.cuda.empty_cache()
    g = torch.Generator(device=dev).manual_seed(1)
    emb = torch.nn.functional.normalize(torch.randn(n, 1024, device=dev, generator=g), dim=-1).half() * 0.2
    kw = dict(beam_size=5, min_gen_len=steps, max_gen_len=(0, steps))
    eng.generate(emb[:8], [3, 256047], beam_size=5, min_gen_len=4, max_gen_len=(0, 4))  # warm-up
    torch.cuda.synchronize()
    ref = None
    for cfg in cfgs:
        for k in KNOBS:
            os.environ.pop(k, None)
        chains = 0
        for item in cfg.split():
            k, v = item.split("=")
            if k == "chains":
                chains = int(v)
            else:
                os.environ[k] = v
        eng.set_chains(chains)
        times = []
        for rep in range(reps + 1):   # rep 0 carries the workspace / KV-cache allocation of a new shape
            torch.cuda.synchronize()
            t0 = time.time()
            toks, lens, scores = eng.generate(emb, [3, 256047], **kw)
            torch.cuda.synchronize()
            times.append((time.time() - t0) * 1e3)
        best = toks[:, 0].cpu()
        sc = scores[:, 0].cpu()
        if ref is None:
            ref = (best, sc)
        same = (best == ref[0]).all(dim=1)
        print(f"{cfg:60s} ms/step " + " ".join(f"{t / (steps + 1):.3f}" for t in times[1:]) +
              f"  (first call {times[0]:.0f} ms)  identical best hypotheses {int(same.sum())}/{n}, "
              f"max |score diff| {(sc - ref[1]).abs().max().item():.2e}", flush=True)


if __name__ == "__main__":
    main()
