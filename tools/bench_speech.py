"""C4 probe: sonar_speech_encoder_eng (fbank + 24-layer conformer + 3-layer attention pooler),
64 clips x 10 s @ 16 kHz, fp16, 1 GPU."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonar_amd.speech_encoder import get_speech_encoder_config, SpeechEncoderEngine, waveform_to_fbank

def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    cfg = get_speech_encoder_config("english")
    d, f = 1024, 4096
    dev = "cuda:0"
    g = torch.Generator(device=dev).manual_seed(1)
    rnd = lambda *s, dt=torch.float16, mean=0.0, std=0.02: (torch.randn(*s, device=dev, generator=g) * std + mean).to(dt)
    f32 = torch.float32
    sd = {"encoder_frontend.post_extract_layer_norm.weight": rnd(160, dt=f32, mean=1.0), "encoder_frontend.post_extract_layer_norm.bias": rnd(160, dt=f32),
          "encoder_frontend.model_dim_proj.weight": rnd(d, 160), "encoder_frontend.model_dim_proj.bias": rnd(d, dt=f32),
          "layer_norm.weight": rnd(d, dt=f32, mean=1.0), "layer_norm.bias": rnd(d, dt=f32),
          "encoder_pooler.decoder_frontend.embed.weight": rnd(1024, d), "encoder_pooler.projection_out.weight": rnd(d, d)}
    for i in range(24):
        p = f"encoder.layers.{i}."
        for ln in ("ffn1_layer_norm", "self_attn_layer_norm", "conv_layer_norm", "ffn2_layer_norm", "layer_norm"):
            sd[p + ln + ".weight"] = rnd(d, dt=f32, mean=1.0); sd[p + ln + ".bias"] = rnd(d, dt=f32)
        for ffn in ("ffn1", "ffn2"):
            sd[p + ffn + ".inner_proj.weight"] = rnd(f, d); sd[p + ffn + ".inner_proj.bias"] = rnd(f, dt=f32)
            sd[p + ffn + ".output_proj.weight"] = rnd(d, f); sd[p + ffn + ".output_proj.bias"] = rnd(d, dt=f32)
        for lin in ("q_proj", "k_proj", "v_proj", "output_proj"):
            sd[p + f"self_attn.{lin}.weight"] = rnd(d, d); sd[p + f"self_attn.{lin}.bias"] = rnd(d, dt=f32)
        sd[p + "self_attn.sdpa.r_proj.weight"] = rnd(d, d)
        sd[p + "self_attn.sdpa.u_bias"] = rnd(16, 64, dt=f32); sd[p + "self_attn.sdpa.v_bias"] = rnd(16, 64, dt=f32)
        sd[p + "conv.pointwise_conv1.weight"] = rnd(2 * d, d, 1); sd[p + "conv.depthwise_conv.weight"] = rnd(d, 1, 31, dt=f32, std=0.1)
        sd[p + "conv.batch_norm.weight"] = rnd(d, dt=f32, mean=1.0); sd[p + "conv.batch_norm.bias"] = rnd(d, dt=f32)
        sd[p + "conv.batch_norm.running_mean"] = rnd(d, dt=f32); sd[p + "conv.batch_norm.running_var"] = rnd(d, dt=f32).abs() + 0.5
        sd[p + "conv.pointwise_conv2.weight"] = rnd(d, d, 1)
    for i in range(3):
        p = f"encoder_pooler.decoder.layers.{i}."
        for att in ("self_attn", "encoder_decoder_attn"):
            for lin in ("q_proj", "k_proj", "v_proj", "output_proj"):
                sd[p + f"{att}.{lin}.weight"] = rnd(d, d); sd[p + f"{att}.{lin}.bias"] = rnd(d, dt=f32)
            sd[p + att + "_layer_norm.weight"] = rnd(d, dt=f32, mean=1.0); sd[p + att + "_layer_norm.bias"] = rnd(d, dt=f32)
        sd[p + "ffn.inner_proj.weight"] = rnd(f, d); sd[p + "ffn.inner_proj.bias"] = rnd(f, dt=f32)
        sd[p + "ffn.output_proj.weight"] = rnd(d, f); sd[p + "ffn.output_proj.bias"] = rnd(d, dt=f32)
        sd[p + "ffn_layer_norm.weight"] = rnd(d, dt=f32, mean=1.0); sd[p + "ffn_layer_norm.bias"] = rnd(d, dt=f32)
    eng = SpeechEncoderEngine(cfg, sd, device=dev)
    del sd
    wavs = torch.rand(n, 160000, device=dev, generator=g) * 2 - 1
    def run():
        feats = torch.stack([waveform_to_fbank(wavs[i]) for i in range(n)])      # [n, 998, 80]
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
          f"{frames} stacked frames; emb finite={bool(torch.isfinite(emb).all())}")

if __name__ == "__main__":
    main()
