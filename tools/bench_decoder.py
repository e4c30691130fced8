"""C5 probe: text_sonar_basic_decoder, beam 5, fp16, batch 256, 64 forced steps (EOS blocked)."""
import sys, time
import torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd.text_decoder import get_text_decoder_config, TextDecoderEngine

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
    cfg = get_text_decoder_config("basic")
    d, f, V = 1024, 8192, cfg.vocab_info.size
    dev = "cuda:0"
    from tools.synth import text_decoder_state_dict
    sd = text_decoder_state_dict(dev)
    g = torch.Generator(device=dev).manual_seed(1)
    eng = TextDecoderEngine(cfg, sd, device=dev)
    del sd
    emb = torch.nn.functional.normalize(torch.randn(n, d, device=dev, generator=g), dim=-1).half() * 0.2
    kw = dict(beam_size=5, min_gen_len=steps, max_gen_len=(0, steps))
    eng.generate(emb[:8], [3, 256047], beam_size=5, min_gen_len=4, max_gen_len=(0, 4))  # warm-up
    torch.cuda.synchronize()
    for rep in range(3):   # rep 0 includes the one-time workspace / KV-cache allocation for this batch size
        t0 = time.time()
        toks, lens, scores = eng.generate(emb, [3, 256047], **kw)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(f"  rep {rep}: {dt*1e3:.1f} ms")
    if len(sys.argv) > 3 and sys.argv[3] == "sample":   # the sampling generator on the same shapes
        from sonar_amd.generation import TopKSampler, TopPSampler
        for smp in (TopPSampler(0.9), TopKSampler(50)):
            for rep in range(2):
                t0 = time.time()
                st, sl, _ = eng.sample(emb, [3, 256047], smp, seed=1, min_gen_len=steps, max_gen_len=(0, steps))
                torch.cuda.synchronize()
                dts = time.time() - t0
            print(f"sampling {smp}: n={n} steps={steps + 1}: {dts*1e3:.1f} ms  {dts/(steps + 1)*1e3:.2f} ms/step  "
                  f"{n/dts:.1f} sentences/s")
    nsteps = steps + 1
    flops = n * 5 * nsteps * (24 * (16 * d * d + 4 * d * f) / 2 * 1 + 2 * d * V)  # self-attn qkv+out (8d^2) + ffn + logits
    print(f"decoder n={n} beam=5 steps={nsteps}: {dt*1e3:.1f} ms  {dt/nsteps*1e3:.2f} ms/step  {n/dt:.1f} sentences/s  lens {lens[0].tolist()}")

if __name__ == "__main__":
    main()
