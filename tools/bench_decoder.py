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
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s, dt=torch.float16, mean=0.0: (torch.randn(*s, device=dev, generator=g) * 0.02 + mean).to(dt)
    sd = {"decoder_frontend.embed.weight": rnd(V, d), "decoder.layer_norm.weight": rnd(d, dt=torch.float32, mean=1.0),
          "decoder.layer_norm.bias": rnd(d, dt=torch.float32)}
    for i in range(24):
        p = f"decoder.layers.{i}."
        for att in ("self_attn", "encoder_decoder_attn"):
            for lin in ("q_proj", "k_proj", "v_proj", "output_proj"):
                sd[p + f"{att}.{lin}.weight"] = rnd(d, d); sd[p + f"{att}.{lin}.bias"] = rnd(d, dt=torch.float32)
        sd[p + "ffn.inner_proj.weight"] = rnd(f, d); sd[p + "ffn.inner_proj.bias"] = rnd(f, dt=torch.float32)
        sd[p + "ffn.output_proj.weight"] = rnd(d, f); sd[p + "ffn.output_proj.bias"] = rnd(d, dt=torch.float32)
        for ln in ("self_attn_layer_norm", "encoder_decoder_attn_layer_norm", "ffn_layer_norm"):
            sd[p + ln + ".weight"] = rnd(d, dt=torch.float32, mean=1.0); sd[p + ln + ".bias"] = rnd(d, dt=torch.float32)
    eng = TextDecoderEngine(cfg, sd, device=dev)
    del sd
    emb = torch.nn.functional.normalize(torch.randn(n, d, device=dev, generator=g), dim=-1).half() * 0.2
    kw = dict(beam_size=5, min_gen_len=steps, max_gen_len=(0, steps))
    eng.generate(emb[:8], [3, 256047], beam_size=5, min_gen_len=4, max_gen_len=(0, 4))  # warm-up
    torch.cuda.synchronize()
    t0 = time.time()
    toks, lens, scores = eng.generate(emb, [3, 256047], **kw)
    torch.cuda.synchronize()
    dt = time.time() - t0
    nsteps = steps + 1
    flops = n * 5 * nsteps * (24 * (16 * d * d + 4 * d * f) / 2 * 1 + 2 * d * V)  # self-attn qkv+out (8d^2) + ffn + logits
    print(f"decoder n={n} beam=5 steps={nsteps}: {dt*1e3:.1f} ms  {dt/nsteps*1e3:.2f} ms/step  {n/dt:.1f} sentences/s  lens {lens[0].tolist()}")

if __name__ == "__main__":
    main()
