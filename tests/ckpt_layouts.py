"""fairseq1-layout checkpoints for the ingest tests: the INVERSE of the reference's key maps, written
out independently (name by name from the fairseq / w2v-BERT module trees) so that a converter bug does
not cancel against itself.

    text encoder   sonar/models/sonar_text/handler.py:52-94
    text decoder   sonar/models/sonar_text/handler.py:122-172
    speech encoder sonar/models/sonar_speech/handler.py:46-110
"""
import re
from typing import Dict

import torch


def _unpermute_control_rows(emb: torch.Tensor) -> torch.Tensor:
    """fairseq2 order (PAD, UNK, BOS, EOS) -> fairseq1 dictionary order (BOS, PAD, EOS, UNK)."""
    out = emb.clone()
    out[[0, 1, 2, 3]] = emb[[2, 0, 3, 1]]
    return out


def text_encoder_to_fairseq1(sd: Dict[str, torch.Tensor]) -> dict:
    out = {}
    for k, v in sd.items():
        m = re.match(r"encoder\.layers\.(\d+)\.(.*)", k)
        if m:
            i, rest = m.groups()
            rest = (rest.replace("self_attn.output_proj.", "self_attn.out_proj.")
                        .replace("ffn.inner_proj.", "fc1.").replace("ffn.output_proj.", "fc2.")
                        .replace("ffn_layer_norm.", "final_layer_norm."))
            out[f"layers.{i}.{rest}"] = v.clone()
        elif k == "encoder_frontend.embed.weight":
            out["embed_tokens.weight"] = _unpermute_control_rows(v)
        elif k.startswith("layer_norm."):
            out[k] = v.clone()
        else:
            raise KeyError(k)
    out["version"] = torch.tensor([3.0])
    out["embed_positions._float_tensor"] = torch.zeros(1)
    return {"state_dict": out, "args": None}


def text_decoder_to_fairseq1(sd: Dict[str, torch.Tensor], tied_storage: bool = True) -> dict:
    out = {}
    for k, v in sd.items():
        m = re.match(r"decoder\.layers\.(\d+)\.(.*)", k)
        if m:
            i, rest = m.groups()
            rest = (rest.replace("encoder_decoder_attn_layer_norm.", "encoder_attn_layer_norm.")
                        .replace("encoder_decoder_attn.output_proj.", "encoder_attn.out_proj.")
                        .replace("encoder_decoder_attn.", "encoder_attn.")
                        .replace("self_attn.output_proj.", "self_attn.out_proj.")
                        .replace("ffn.inner_proj.", "fc1.").replace("ffn.output_proj.", "fc2.")
                        .replace("ffn_layer_norm.", "final_layer_norm."))
            out[f"layers.{i}.{rest}"] = v.clone()
        elif k == "decoder_frontend.embed.weight":
            out["embed_tokens.weight"] = _unpermute_control_rows(v)
        elif k.startswith("decoder.layer_norm."):
            out[k[len("decoder."):]] = v.clone()
        else:
            raise KeyError(k)
    # share_decoder_input_output_embed: fairseq saves the tied projection under its own key; in the
    # file both keys alias one storage (torch.save keeps the aliasing)
    out["output_projection.weight"] = out["embed_tokens.weight"] if tied_storage else out["embed_tokens.weight"].clone()
    out["version"] = torch.tensor([3.0])
    out["embed_positions._float_tensor"] = torch.zeros(1)
    return {"state_dict": out}


def speech_encoder_to_fairseq1(sd: Dict[str, torch.Tensor]) -> dict:
    out = {}
    W = "encoder.w2v_model."
    for k, v in sd.items():
        v = v.clone()
        m = re.match(r"encoder\.layers\.(\d+)\.(.*)", k)
        p = re.match(r"encoder_pooler\.decoder\.layers\.(\d+)\.(.*)", k)
        if m:
            i, rest = m.groups()
            L = f"{W}encoder.layers.{i}."
            table = [
                (r"^ffn(1|2)_layer_norm\.", r"ffn\1.layer_norm."),
                (r"^ffn(1|2)\.inner_proj\.", r"ffn\1.w_1."),
                (r"^ffn(1|2)\.output_proj\.", r"ffn\1.w_2."),
                (r"^self_attn\.q_proj\.", "self_attn.linear_q."),
                (r"^self_attn\.k_proj\.", "self_attn.linear_k."),
                (r"^self_attn\.v_proj\.", "self_attn.linear_v."),
                (r"^self_attn\.output_proj\.", "self_attn.linear_out."),
                (r"^self_attn\.sdpa\.r_proj\.", "self_attn.linear_pos."),
                (r"^self_attn\.sdpa\.u_bias$", "self_attn.pos_bias_u"),
                (r"^self_attn\.sdpa\.v_bias$", "self_attn.pos_bias_v"),
                (r"^conv_layer_norm\.", "conv_module.layer_norm."),
                (r"^conv\.(pointwise_conv1|pointwise_conv2|depthwise_conv|batch_norm)\.", r"conv_module.\1."),
                (r"^layer_norm\.", "final_layer_norm."),
                (r"^self_attn_layer_norm\.", "self_attn_layer_norm."),
            ]
            for pat, rep in table:
                new, n = re.subn(pat, rep, rest)
                if n:
                    out[L + new] = v
                    break
            else:
                raise KeyError(k)
        elif p:
            i, rest = p.groups()
            rest = (rest.replace("encoder_decoder_attn_layer_norm.", "encoder_attn_layer_norm.")
                        .replace("encoder_decoder_attn.output_proj.", "encoder_attn.out_proj.")
                        .replace("encoder_decoder_attn.", "encoder_attn.")
                        .replace("self_attn.output_proj.", "self_attn.out_proj.")
                        .replace("ffn.inner_proj.", "fc1.").replace("ffn.output_proj.", "fc2.")
                        .replace("ffn_layer_norm.", "final_layer_norm."))
            out[f"decoder.layers.{i}.{rest}"] = v
        elif k.startswith("encoder_frontend.post_extract_layer_norm."):
            out[W + "layer_norm." + k.rsplit(".", 1)[1]] = v
        elif k.startswith("encoder_frontend.model_dim_proj."):
            out[W + "post_extract_proj." + k.rsplit(".", 1)[1]] = v
        elif k.startswith("layer_norm."):
            # the redundant post-conformer LayerNorm of the fairseq model (handler.py:102-108)
            out[W + "encoder.layer_norm." + k.rsplit(".", 1)[1]] = v
        elif k == "encoder_pooler.decoder_frontend.embed.weight":
            out["decoder.embed_tokens.weight"] = v
        elif k == "encoder_pooler.projection_out.weight":
            out["decoder.embed_out"] = v
        else:
            raise KeyError(k)
    d = sd["layer_norm.weight"].shape[0]
    # pre-training leftovers the handler deletes (handler.py:55-61)
    out[W + "mask_emb"] = torch.randn(d)
    out[W + "encoder.pos_conv.0.bias"] = torch.randn(d)
    out[W + "encoder.pos_conv.0.weight_g"] = torch.randn(1, 1, 8)
    out[W + "encoder.pos_conv.0.weight_v"] = torch.randn(d, 4, 8)
    for key in [k for k in out if k.endswith("batch_norm.running_var")]:
        out[key.replace("running_var", "num_batches_tracked")] = torch.tensor(7)
    return {"model": out, "cfg": None}
