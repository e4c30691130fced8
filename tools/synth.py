"""Random-init state dicts of the three SONAR architectures, generated on the GPU in fp16
(there is no network for checkpoints): shared by bench.py and the tools/ probes."""
import torch

BATCH, SEQ = 1024, 128
D, F, L, H, V = 1024, 8192, 24, 16, 256206


def text_encoder_state_dict(device, seed=1234):
    """Random-init weights of the `basic` architecture, generated on the GPU in fp16
    (Linear/Embedding ~ N(0, 0.02^2), LN weight 1 + N(0, 0.02^2))."""
    import torch

    g = torch.Generator(device=device).manual_seed(seed)

    def rnd(*shape, dtype=torch.float16, mean=0.0):
        return (torch.randn(*shape, device=device, generator=g) * 0.02 + mean).to(dtype)

    sd = {"encoder_frontend.embed.weight": rnd(V, D),
          "layer_norm.weight": rnd(D, dtype=torch.float32, mean=1.0),
          "layer_norm.bias": rnd(D, dtype=torch.float32)}
    for i in range(L):
        p = f"encoder.layers.{i}."
        for name, shape in (("self_attn.q_proj", (D, D)), ("self_attn.k_proj", (D, D)),
                            ("self_attn.v_proj", (D, D)), ("self_attn.output_proj", (D, D)),
                            ("ffn.inner_proj", (F, D)), ("ffn.output_proj", (D, F))):
            sd[p + name + ".weight"] = rnd(*shape)
            sd[p + name + ".bias"] = rnd(shape[0], dtype=torch.float32)
        for name in ("self_attn_layer_norm", "ffn_layer_norm"):
            sd[p + name + ".weight"] = rnd(D, dtype=torch.float32, mean=1.0)
            sd[p + name + ".bias"] = rnd(D, dtype=torch.float32)
    return sd




def text_decoder_state_dict(device, seed=1):
    d, f = D, F
    g = torch.Generator(device=device).manual_seed(seed)
    rnd = lambda *s, dt=torch.float16, mean=0.0: (torch.randn(*s, device=device, generator=g) * 0.02 + mean).to(dt)
    sd = {"decoder_frontend.embed.weight": rnd(V, d), "decoder.layer_norm.weight": rnd(d, dt=torch.float32, mean=1.0),
          "decoder.layer_norm.bias": rnd(d, dt=torch.float32)}
    for i in range(24):
        p = f"decoder.layers.{i}."
        for att in ("self_attn", "encoder_decoder_attn"):
            for lin in ("q_proj", "k_proj", "v_proj", "output_proj"):
                sd[p + f"{att}.{lin}.weight"] = rnd(d, d)
                sd[p + f"{att}.{lin}.bias"] = rnd(d, dt=torch.float32)
        sd[p + "ffn.inner_proj.weight"] = rnd(f, d)
        sd[p + "ffn.inner_proj.bias"] = rnd(f, dt=torch.float32)
        sd[p + "ffn.output_proj.weight"] = rnd(d, f)
        sd[p + "ffn.output_proj.bias"] = rnd(d, dt=torch.float32)
        for ln in ("self_attn_layer_norm", "encoder_decoder_attn_layer_norm", "ffn_layer_norm"):
            sd[p + ln + ".weight"] = rnd(d, dt=torch.float32, mean=1.0)
            sd[p + ln + ".bias"] = rnd(d, dt=torch.float32)
    return sd


def speech_encoder_state_dict(device, seed=1, pooler_layers=3):
    d, f = 1024, 4096
    g = torch.Generator(device=device).manual_seed(seed)
    f32 = torch.float32
    rnd = lambda *s, dt=torch.float16, mean=0.0, std=0.02: (torch.randn(*s, device=device, generator=g) * std + mean).to(dt)
    sd = {"encoder_frontend.post_extract_layer_norm.weight": rnd(160, dt=f32, mean=1.0),
          "encoder_frontend.post_extract_layer_norm.bias": rnd(160, dt=f32),
          "encoder_frontend.model_dim_proj.weight": rnd(d, 160), "encoder_frontend.model_dim_proj.bias": rnd(d, dt=f32),
          "layer_norm.weight": rnd(d, dt=f32, mean=1.0), "layer_norm.bias": rnd(d, dt=f32),
          "encoder_pooler.decoder_frontend.embed.weight": rnd(1024, d), "encoder_pooler.projection_out.weight": rnd(d, d)}
    for i in range(24):
        p = f"encoder.layers.{i}."
        for ln in ("ffn1_layer_norm", "self_attn_layer_norm", "conv_layer_norm", "ffn2_layer_norm", "layer_norm"):
            sd[p + ln + ".weight"] = rnd(d, dt=f32, mean=1.0)
            sd[p + ln + ".bias"] = rnd(d, dt=f32)
        for ffn in ("ffn1", "ffn2"):
            sd[p + ffn + ".inner_proj.weight"] = rnd(f, d)
            sd[p + ffn + ".inner_proj.bias"] = rnd(f, dt=f32)
            sd[p + ffn + ".output_proj.weight"] = rnd(d, f)
            sd[p + ffn + ".output_proj.bias"] = rnd(d, dt=f32)
        for lin in ("q_proj", "k_proj", "v_proj", "output_proj"):
            sd[p + f"self_attn.{lin}.weight"] = rnd(d, d)
            sd[p + f"self_attn.{lin}.bias"] = rnd(d, dt=f32)
        sd[p + "self_attn.sdpa.r_proj.weight"] = rnd(d, d)
        sd[p + "self_attn.sdpa.u_bias"] = rnd(16, 64, dt=f32)
        sd[p + "self_attn.sdpa.v_bias"] = rnd(16, 64, dt=f32)
        sd[p + "conv.pointwise_conv1.weight"] = rnd(2 * d, d, 1)
        sd[p + "conv.depthwise_conv.weight"] = rnd(d, 1, 31, dt=f32, std=0.1)
        sd[p + "conv.batch_norm.weight"] = rnd(d, dt=f32, mean=1.0)
        sd[p + "conv.batch_norm.bias"] = rnd(d, dt=f32)
        sd[p + "conv.batch_norm.running_mean"] = rnd(d, dt=f32)
        sd[p + "conv.batch_norm.running_var"] = rnd(d, dt=f32).abs() + 0.5
        sd[p + "conv.pointwise_conv2.weight"] = rnd(d, d, 1)
    for i in range(pooler_layers):
        p = f"encoder_pooler.decoder.layers.{i}."
        for att in ("self_attn", "encoder_decoder_attn"):
            for lin in ("q_proj", "k_proj", "v_proj", "output_proj"):
                sd[p + f"{att}.{lin}.weight"] = rnd(d, d)
                sd[p + f"{att}.{lin}.bias"] = rnd(d, dt=f32)
            sd[p + att + "_layer_norm.weight"] = rnd(d, dt=f32, mean=1.0)
            sd[p + att + "_layer_norm.bias"] = rnd(d, dt=f32)
        sd[p + "ffn.inner_proj.weight"] = rnd(f, d)
        sd[p + "ffn.inner_proj.bias"] = rnd(f, dt=f32)
        sd[p + "ffn.output_proj.weight"] = rnd(d, f)
        sd[p + "ffn.output_proj.bias"] = rnd(d, dt=f32)
        sd[p + "ffn_layer_norm.weight"] = rnd(d, dt=f32, mean=1.0)
        sd[p + "ffn_layer_norm.bias"] = rnd(d, dt=f32)
    return sd
