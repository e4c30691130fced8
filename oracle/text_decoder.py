"""ORACLE -- test infrastructure only.  Never imported by the product package.

CPU / plain-PyTorch fp32 restatement of the reference's EmbeddingToText hot path
(facebookresearch/SONAR v0.4.0): the `text_sonar_basic_decoder` forward pass and
the beam search that `EmbeddingToTextModelPipeline.predict` drives
(sonar/inference_pipelines/text.py:305-346).

Pinning status:
  * the decoder stack (pre-LN self-attn / cross-attn / FFN layers, final LN, tied
    output projection; sonar/models/sonar_text/factory.py:229-315) is pinned against
    HuggingFace `M2M100Decoder` through tests/golden/m2m100_decoder_twin.pt
    (generator tests/golden/make_golden_decoder.py);
  * the reference's own golden logits (tests/integration_tests/test_text_sonar.py:61-105)
    and translations (:107-118) need the real checkpoint: PARITY UNPINNED for those;
  * the beam search restates fairseq2 ~=0.4 `BeamSearchSeq2SeqGenerator` /
    `StandardBeamSearchAlgorithm` (un-vendored dependency, pyproject.toml:27) from its
    published behaviour.  The beam_size=1 generation loop (prompt forcing,
    incremental steps, EOS stop, min-length EOS suppression) is pinned token-for-token to
    HuggingFace `generate(num_beams=1)` through tests/golden/m2m100_greedy_twin.pt
    (generator tests/golden/make_golden_generate.py); beam_size>1 hypothesis bookkeeping
    (2*beam candidates, length normalisation, forced EOS at the cap) is PARITY UNPINNED
    beyond self-consistency (scores == teacher-forced log-probs, ordering, EOS rules) --
    HF's beam scorer finalises hypotheses differently and cannot serve as its twin;
  * the sampling generator (second half of this file) restates fairseq2's TopKSampler /
    TopPSampler / SamplingSeq2SeqGenerator from recall: PARITY UNPINNED.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

from .text_encoder import sinusoidal_table


@dataclass
class OracleTextDecoderConfig:
    """Forward-relevant fields of SonarTextDecoderConfig (config.py:130-190; `basic` :197-219)."""

    model_dim: int = 1024
    num_layers: int = 24
    num_heads: int = 16
    ffn_inner_dim: int = 8192
    vocab_size: int = 256206
    max_seq_len: int = 512
    pad_idx: int = 1            # model vocab_info.pad_idx -> position offset pad_idx + 1
    input_dim: Optional[int] = None  # dimension of the conditioning vector (default model_dim)
    no_scale_embedding: bool = False
    ln_eps: float = 1e-5

    @property
    def pos_offset(self) -> int:
        return self.pad_idx + 1

    @property
    def cond_dim(self) -> int:
        return self.input_dim or self.model_dim


def param_names(cfg: OracleTextDecoderConfig) -> List[str]:
    """fairseq2-style names after sonar/models/sonar_text/handler.py:139-159."""
    names = ["decoder_frontend.embed.weight", "decoder.layer_norm.weight", "decoder.layer_norm.bias"]
    for i in range(cfg.num_layers):
        p = f"decoder.layers.{i}."
        for att in ("self_attn", "encoder_decoder_attn"):
            for lin in ("q_proj", "k_proj", "v_proj", "output_proj"):
                names += [p + f"{att}.{lin}.weight", p + f"{att}.{lin}.bias"]
        for lin in ("ffn.inner_proj", "ffn.output_proj"):
            names += [p + lin + ".weight", p + lin + ".bias"]
        for ln in ("self_attn_layer_norm", "encoder_decoder_attn_layer_norm", "ffn_layer_norm"):
            names += [p + ln + ".weight", p + ln + ".bias"]
    return names


def param_shape(cfg: OracleTextDecoderConfig, name: str) -> Tuple[int, ...]:
    d, f, c = cfg.model_dim, cfg.ffn_inner_dim, cfg.cond_dim
    if name == "decoder_frontend.embed.weight":
        return (cfg.vocab_size, d)
    if "layer_norm" in name:
        return (d,)
    w = name.endswith("weight")
    if "ffn.inner_proj" in name:
        return (f, d) if w else (f,)
    if "ffn.output_proj" in name:
        return (d, f) if w else (d,)
    if "encoder_decoder_attn.k_proj" in name or "encoder_decoder_attn.v_proj" in name:
        return (d, c) if w else (d,)
    return (d, d) if w else (d,)


def make_synthetic_params(cfg: OracleTextDecoderConfig, seed: int = 4321, std: float = 0.02) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    out = {}
    for name in param_names(cfg):
        t = torch.randn(param_shape(cfg, name), generator=g, dtype=torch.float32) * std
        if "layer_norm" in name and name.endswith("weight"):
            t = t + 1.0
        out[name] = t
    return out


def _ln(x, w, b, eps):
    return F.layer_norm(x, (x.shape[-1],), w, b, eps)


def _mha(params, prefix, q_in, kv_in, heads, causal):
    d = q_in.shape[-1]
    n, tq = q_in.shape[:2]
    tk = kv_in.shape[1]
    dh = d // heads
    q = F.linear(q_in, params[prefix + "q_proj.weight"], params[prefix + "q_proj.bias"])
    k = F.linear(kv_in, params[prefix + "k_proj.weight"], params[prefix + "k_proj.bias"])
    v = F.linear(kv_in, params[prefix + "v_proj.weight"], params[prefix + "v_proj.bias"])
    q = q.view(n, tq, heads, dh).transpose(1, 2)
    k = k.view(n, tk, heads, dh).transpose(1, 2)
    v = v.view(n, tk, heads, dh).transpose(1, 2)
    att = torch.matmul(q, k.transpose(-1, -2)) * dh ** -0.5
    if causal:
        mask = torch.triu(torch.ones(tq, tk, dtype=torch.bool), diagonal=1)
        att = att.masked_fill(mask, -torch.inf)
    att = torch.softmax(att, dim=-1)
    y = torch.matmul(att, v).transpose(1, 2).reshape(n, tq, d)
    return F.linear(y, params[prefix + "output_proj.weight"], params[prefix + "output_proj.bias"])


@torch.inference_mode()
def decoder_logits(params: Dict[str, torch.Tensor], cfg: OracleTextDecoderConfig,
                   embeddings: torch.Tensor, prev_tokens: torch.Tensor) -> torch.Tensor:
    """Teacher-forced logits [N, T, V] for decoder inputs `prev_tokens` [N, T], conditioned on the
    sentence embeddings [N, cond_dim] used as a length-1 encoder output
    (SonarEncoderDecoderModel.encode/decode/project, sonar/models/sonar_translation/model.py:48-78;
    ConditionalTransformerDecoderModel, sonar/nn/conditional_decoder_model.py:66-94).
    Same call as the reference test (test_text_sonar.py:61-105, prev tokens [[3, 333]])."""
    d = cfg.model_dim
    n, t = prev_tokens.shape
    scale = 1.0 if cfg.no_scale_embedding else math.sqrt(d)
    E = params["decoder_frontend.embed.weight"]
    x = E[prev_tokens].float() * scale
    x = x + sinusoidal_table(cfg.pos_offset + t, d)[cfg.pos_offset:].unsqueeze(0)
    enc = embeddings.float().unsqueeze(1)  # [N, 1, cond]
    for i in range(cfg.num_layers):
        p = f"decoder.layers.{i}."
        # StandardTransformerDecoderLayer, norm_order PRE (factory.py:261-274)
        h = _ln(x, params[p + "self_attn_layer_norm.weight"], params[p + "self_attn_layer_norm.bias"], cfg.ln_eps)
        x = x + _mha(params, p + "self_attn.", h, h, cfg.num_heads, causal=True)
        h = _ln(x, params[p + "encoder_decoder_attn_layer_norm.weight"],
                params[p + "encoder_decoder_attn_layer_norm.bias"], cfg.ln_eps)
        x = x + _mha(params, p + "encoder_decoder_attn.", h, enc, cfg.num_heads, causal=False)
        h = _ln(x, params[p + "ffn_layer_norm.weight"], params[p + "ffn_layer_norm.bias"], cfg.ln_eps)
        h = F.relu(F.linear(h, params[p + "ffn.inner_proj.weight"], params[p + "ffn.inner_proj.bias"]))
        x = x + F.linear(h, params[p + "ffn.output_proj.weight"], params[p + "ffn.output_proj.bias"])
    # StandardTransformerDecoder(norm_order=PRE) => final LayerNorm (factory.py:294-301)
    x = _ln(x, params["decoder.layer_norm.weight"], params["decoder.layer_norm.bias"], cfg.ln_eps)
    return F.linear(x, E)  # TiedProjection(embed.weight, bias=None) (factory.py:306-307)


@dataclass
class Hypothesis:
    seq: torch.Tensor          # generated tokens after the prompt, including the final EOS
    score: float               # cumulative log-prob, length-normalised when normalize_scores
    step_scores: torch.Tensor  # per-step log-probs of `seq`


@torch.inference_mode()
def beam_search(params, cfg: OracleTextDecoderConfig, embeddings: torch.Tensor, prompt: Sequence[int],
                beam_size: int = 5, min_gen_len: int = 1, max_gen_len: Tuple[int, int] = (1, 128),
                max_seq_len: Optional[int] = None, normalize_scores: bool = True, len_penalty: float = 1.0,
                unk_penalty: float = 0.0, temperature: float = 1.0, pad_idx: int = 0, unk_idx: int = 1,
                eos_idx: int = 3, source_len: Optional[int] = None) -> List[List[Hypothesis]]:
    """fairseq2 BeamSearchSeq2SeqGenerator defaults for a batch of sentence
    embeddings; one independent beam per embedding (the reference processes the beams of a
    batch jointly, which is arithmetically the same).  Returns, per embedding, its finished
    hypotheses sorted best first (the pipeline decodes hypotheses[0].seq)."""
    model_max = max_seq_len if max_seq_len is not None else cfg.max_seq_len
    plen = len(prompt)
    # fairseq2: max_gen_len = a * max_source_len + b with max_source_len = source_seqs.size(1).  The
    # vec2text pipeline passes the stacked embeddings [n, model_dim] as source_seqs (text.py:329-333),
    # so the "source length" is model_dim there; text / speech sources pass their token / frame count.
    if source_len is None:
        source_len = cfg.cond_dim
    gen_cap = int(max_gen_len[0] * source_len + max_gen_len[1])
    max_len = min(plen + gen_cap, model_max)
    min_len = min(plen + min_gen_len, max_len)
    results: List[List[Hypothesis]] = []
    for e in embeddings:
        emb = e.unsqueeze(0)
        seqs = torch.tensor([list(prompt)], dtype=torch.int64)          # [beams, step_nr]
        cum = torch.zeros(1, plen, dtype=torch.float32)                 # cumulative scores per position
        # prefill: the score of every prompt token after the first is accumulated (teacher forced)
        if plen > 1:
            lp = torch.log_softmax(decoder_logits(params, cfg, emb, seqs[:, :-1]) / temperature, dim=-1, dtype=torch.float32)
            ps = lp[0, torch.arange(plen - 1), seqs[0, 1:]].cumsum(0)
            cum[0, 1:] = ps
        finished: List[Hypothesis] = []
        step_nr = plen
        while True:
            b = seqs.shape[0]
            logits = decoder_logits(params, cfg, emb.expand(b, -1), seqs)[:, -1]
            lprobs = torch.log_softmax(logits / temperature, dim=-1, dtype=torch.float32)
            if step_nr == max_len - 1:
                lprobs[:, :eos_idx] = -torch.inf
                lprobs[:, eos_idx + 1:] = -torch.inf
            else:
                lprobs[:, unk_idx] -= unk_penalty
                lprobs[:, pad_idx] = -torch.inf
                if step_nr < min_len:
                    lprobs[:, eos_idx] = -torch.inf
            v = lprobs.shape[1]
            cand = (lprobs + cum[:, -1:]).view(-1)
            k = min(2 * beam_size, v - 1)
            top_scores, top_idx = torch.topk(cand, k)
            seq_idx, vocab_idx = top_idx // v, top_idx % v
            eos_mask = vocab_idx == eos_idx
            done = False
            head = eos_mask[:beam_size]
            for si, sc in zip(seq_idx[:beam_size][head].tolist(), top_scores[:beam_size][head].tolist()):
                seq = torch.cat([seqs[si], torch.tensor([eos_idx])])
                steps = torch.cat([cum[si], torch.tensor([sc])])
                seq_len = step_nr + 1
                out_steps = steps[plen:seq_len].clone()
                prev = steps[plen - 1:seq_len - 1]
                out_steps = out_steps - prev
                score = sc / (seq_len - 1) ** len_penalty if normalize_scores else sc
                finished.append(Hypothesis(seq[plen:], float(score), out_steps))
                if len(finished) == beam_size:
                    done = True
                    break
            if done:
                break
            keep = ~eos_mask
            seq_idx, vocab_idx, top_scores = seq_idx[keep][:beam_size], vocab_idx[keep][:beam_size], top_scores[keep][:beam_size]
            seqs = torch.cat([seqs[seq_idx], vocab_idx.unsqueeze(1)], dim=1)
            cum = torch.cat([cum[seq_idx], top_scores.unsqueeze(1)], dim=1)
            step_nr += 1
            if step_nr >= max_len:  # cannot happen: the step before forces EOS
                break
        finished.sort(key=lambda h: h.score, reverse=True)
        results.append(finished)
    return results


class _IncrementalState:
    """Per-layer self-attention K / V of the positions decoded so far, [beams, t, d] (fairseq2's IncrementalStateBag)."""

    def __init__(self, num_layers: int):
        self.k: List[Optional[torch.Tensor]] = [None] * num_layers
        self.v: List[Optional[torch.Tensor]] = [None] * num_layers

    def reorder(self, idx: torch.Tensor) -> None:
        for i in range(len(self.k)):
            if self.k[i] is not None:
                self.k[i] = self.k[i].index_select(0, idx)
                self.v[i] = self.v[i].index_select(0, idx)


def _decoder_step_cached(params, cfg: OracleTextDecoderConfig, emb: torch.Tensor, tokens: torch.Tensor, pos: int,
                         st: _IncrementalState) -> torch.Tensor:
    """Logits [beams, V] of ONE new position `pos` for `tokens` [beams], given the cached prefix: the same arithmetic as
    decoder_logits(...)[:, -1], evaluated the way the reference evaluates it (incremental decoding with a K/V cache,
    cache rows re-ordered by index_select after each beam step).  Used for the CPU baseline timing of the C5 leg
    (bench.py) and checked against the quadratic restatement in tests/test_oracle_decoder_cpu.py."""
    d, hn = cfg.model_dim, cfg.num_heads
    dh = d // hn
    b = tokens.shape[0]
    scale = 1.0 if cfg.no_scale_embedding else math.sqrt(d)
    E = params["decoder_frontend.embed.weight"]
    x = E[tokens].float() * scale + sinusoidal_table(cfg.pos_offset + pos + 1, d)[cfg.pos_offset + pos]
    enc = emb.float().unsqueeze(1)
    for i in range(cfg.num_layers):
        p = f"decoder.layers.{i}."
        h = _ln(x, params[p + "self_attn_layer_norm.weight"], params[p + "self_attn_layer_norm.bias"], cfg.ln_eps)
        q = F.linear(h, params[p + "self_attn.q_proj.weight"], params[p + "self_attn.q_proj.bias"])
        k = F.linear(h, params[p + "self_attn.k_proj.weight"], params[p + "self_attn.k_proj.bias"]).unsqueeze(1)
        v = F.linear(h, params[p + "self_attn.v_proj.weight"], params[p + "self_attn.v_proj.bias"]).unsqueeze(1)
        st.k[i] = k if st.k[i] is None else torch.cat([st.k[i], k], dim=1)
        st.v[i] = v if st.v[i] is None else torch.cat([st.v[i], v], dim=1)
        t = st.k[i].shape[1]
        qh = q.view(b, 1, hn, dh).transpose(1, 2)
        kh = st.k[i].view(b, t, hn, dh).transpose(1, 2)
        vh = st.v[i].view(b, t, hn, dh).transpose(1, 2)
        att = torch.softmax(torch.matmul(qh, kh.transpose(-1, -2)) * dh ** -0.5, dim=-1)
        y = torch.matmul(att, vh).transpose(1, 2).reshape(b, d)
        x = x + F.linear(y, params[p + "self_attn.output_proj.weight"], params[p + "self_attn.output_proj.bias"])
        h = _ln(x, params[p + "encoder_decoder_attn_layer_norm.weight"],
                params[p + "encoder_decoder_attn_layer_norm.bias"], cfg.ln_eps)
        x = x + _mha(params, p + "encoder_decoder_attn.", h.unsqueeze(1), enc.expand(b, -1, -1), hn, causal=False).squeeze(1)
        h = _ln(x, params[p + "ffn_layer_norm.weight"], params[p + "ffn_layer_norm.bias"], cfg.ln_eps)
        h = F.relu(F.linear(h, params[p + "ffn.inner_proj.weight"], params[p + "ffn.inner_proj.bias"]))
        x = x + F.linear(h, params[p + "ffn.output_proj.weight"], params[p + "ffn.output_proj.bias"])
    x = _ln(x, params["decoder.layer_norm.weight"], params["decoder.layer_norm.bias"], cfg.ln_eps)
    return F.linear(x, E)


@torch.inference_mode()
def beam_search_incremental(params, cfg: OracleTextDecoderConfig, embeddings: torch.Tensor, prompt: Sequence[int],
                            beam_size: int = 5, min_gen_len: int = 1, max_gen_len: Tuple[int, int] = (1, 128),
                            max_seq_len: Optional[int] = None, normalize_scores: bool = True, len_penalty: float = 1.0,
                            pad_idx: int = 0, eos_idx: int = 3, source_len: Optional[int] = None,
                            margins_out: Optional[list] = None) -> List[List[Hypothesis]]:
    """beam_search() with incremental decoding (K/V cache, index_select re-ordering): the evaluation order of the
    reference's generator.  Same hypotheses as beam_search (tests/test_oracle_decoder_cpu.py); linear instead of
    quadratic work in the output length, so this is the variant bench.py times as the CPU baseline.

    margins_out (a list): per sentence (decision margin, final margin) of THIS search is appended --
    decision margin = over all steps, the smallest gap between neighbours of the sorted candidate list among the candidates
    the beam rules consumed plus the first one they did not (a gap below the arithmetic noise of another implementation
    means that implementation may legitimately order the two candidates the other way round); final margin = score gap
    between the best finished hypothesis and the runner-up; third entry = the decision margin without the step at the
    length cap (every candidate a forced EOS: the engine ranks those by the final scores and reports that gap as its final
    margin) -- the number the engine's own decision margin is cross-checked against.  The parity tests excuse a token
    mismatch ONLY by these oracle-side numbers (VERDICT r5 "weak" 2), never by the engine's own report."""
    model_max = max_seq_len if max_seq_len is not None else cfg.max_seq_len
    plen = len(prompt)
    if source_len is None:
        source_len = cfg.cond_dim
    max_len = min(plen + int(max_gen_len[0] * source_len + max_gen_len[1]), model_max)
    min_len = min(plen + min_gen_len, max_len)
    results: List[List[Hypothesis]] = []
    for e in embeddings:
        emb = e.unsqueeze(0)
        st = _IncrementalState(cfg.num_layers)
        seqs = torch.tensor([list(prompt)], dtype=torch.int64)
        cum = torch.zeros(1, plen, dtype=torch.float32)
        for pos in range(plen - 1):      # prefill: teacher-forced prompt, scores of prompt tokens 1.. accumulate
            lp = torch.log_softmax(_decoder_step_cached(params, cfg, emb, seqs[:, pos], pos, st), dim=-1, dtype=torch.float32)
            cum[0, pos + 1] = cum[0, pos] + lp[0, seqs[0, pos + 1]]
        finished: List[Hypothesis] = []
        step_nr = plen
        dec_margin = dec_margin_free = float("inf")
        while True:
            b = seqs.shape[0]
            logits = _decoder_step_cached(params, cfg, emb.expand(b, -1), seqs[:, -1], step_nr - 1, st)
            lprobs = torch.log_softmax(logits, dim=-1, dtype=torch.float32)
            if step_nr == max_len - 1:
                lprobs[:, :eos_idx] = -torch.inf
                lprobs[:, eos_idx + 1:] = -torch.inf
            else:
                lprobs[:, pad_idx] = -torch.inf
                if step_nr < min_len:
                    lprobs[:, eos_idx] = -torch.inf
            v = lprobs.shape[1]
            cand = (lprobs + cum[:, -1:]).view(-1)
            top_scores, top_idx = torch.topk(cand, min(2 * beam_size, v - 1))
            seq_idx, vocab_idx = top_idx // v, top_idx % v
            eos_mask = vocab_idx == eos_idx
            done = False
            head = eos_mask[:beam_size]
            completing = -1          # sorted index of the EOS candidate that completes the beam in this step
            head_pos = torch.nonzero(head).view(-1).tolist()
            for hp, si, sc in zip(head_pos, seq_idx[:beam_size][head].tolist(), top_scores[:beam_size][head].tolist()):
                seq = torch.cat([seqs[si], torch.tensor([eos_idx])])
                steps = torch.cat([cum[si], torch.tensor([sc])])
                seq_len = step_nr + 1
                out_steps = steps[plen:seq_len] - steps[plen - 1:seq_len - 1]
                score = sc / (seq_len - 1) ** len_penalty if normalize_scores else sc
                finished.append(Hypothesis(seq[plen:], float(score), out_steps))
                if len(finished) == beam_size:
                    done = True
                    completing = hp
                    break
            if margins_out is not None:
                # candidates the beam rules consumed (csrc/decoder.hip states the same rule for the engine's own report): up to
                # the EOS candidate that completed the beam, else up to the beam_size-th non-EOS candidate; + the first
                # candidate NOT consumed
                last = completing
                if not done:
                    non_eos = torch.nonzero(~eos_mask).view(-1)
                    last = int(non_eos[min(beam_size, non_eos.numel()) - 1]) if non_eos.numel() else top_scores.numel() - 1
                upto = min(last + 1, top_scores.numel() - 1)
                if upto >= 1:
                    gaps = top_scores[:upto] - top_scores[1:upto + 1]
                    gaps = gaps[torch.isfinite(gaps)]
                    if gaps.numel():
                        dec_margin = min(dec_margin, float(gaps.min()))
                        if step_nr != max_len - 1:   # not the step at the length cap, where every candidate is a forced EOS
                            dec_margin_free = min(dec_margin_free, float(gaps.min()))
            if done:
                break
            keep = ~eos_mask
            seq_idx, vocab_idx, top_scores = seq_idx[keep][:beam_size], vocab_idx[keep][:beam_size], top_scores[keep][:beam_size]
            seqs = torch.cat([seqs[seq_idx], vocab_idx.unsqueeze(1)], dim=1)
            cum = torch.cat([cum[seq_idx], top_scores.unsqueeze(1)], dim=1)
            st.reorder(seq_idx)
            step_nr += 1
            if step_nr >= max_len:
                break
        finished.sort(key=lambda h: h.score, reverse=True)
        results.append(finished)
        if margins_out is not None:
            fin = finished[0].score - finished[1].score if len(finished) > 1 else float("inf")
            margins_out.append((dec_margin, fin, dec_margin_free))
    return results


@torch.inference_mode()
def greedy_decode(params, cfg, embeddings, prompt, max_new: int, eos_idx: int = 3, pad_idx: int = 0):
    """Plain argmax decoding (used to sanity-check beam_size=1)."""
    outs = []
    for e in embeddings:
        seq = list(prompt)
        for i in range(max_new):
            lg = decoder_logits(params, cfg, e.unsqueeze(0), torch.tensor([seq]))[0, -1].clone()
            lg[pad_idx] = -torch.inf
            if i == 0:
                lg[eos_idx] = -torch.inf  # min_gen_len = 1
            tok = int(lg.argmax())
            seq.append(tok)
            if tok == eos_idx:
                break
        outs.append(seq[len(prompt):])
    return outs


# ------------------------------------------------------------------------------ sampling
# fairseq2 ~=0.4 SamplingSeq2SeqGenerator / TopKSampler / TopPSampler (un-vendored; restated from
# their published behaviour, sonar/inference_pipelines/text.py:315-320 is the call site):
# PARITY UNPINNED -- no sampling test or fixture exists in the reference, and a sampled sequence
# depends on the random stream.  What the HIP path is held to: the same kept set as this restatement
# on the same logits, the same token for the same random word, top_k=1 == greedy, scores == the
# teacher-forced log-probs of the sampled tokens.
SAMPLE_THREADS = 1024          # workgroup size of csrc/sampling.hip: fixes the order the draw walks in
_MASK64 = (1 << 64) - 1


def sampling_probs(logits: torch.Tensor, temperature: float = 1.0, pad_idx: int = 0, eos_idx: int = 3,
                   block_eos: bool = False, unk_idx: int = 1, unk_penalty: float = 0.0) -> torch.Tensor:
    """probs = softmax(logits / T, fp32); pad (and EOS before min_len) zeroed, probs[unk] -= unk_penalty
    (fs2-recall; held at >= 0: a negative probability would make the reference's multinomial raise), NOT renormalised."""
    probs = torch.softmax(logits.float() / temperature, dim=-1, dtype=torch.float32).clone()
    probs[..., pad_idx] = 0.0
    if block_eos:
        probs[..., eos_idx] = 0.0
    if unk_penalty != 0.0:
        probs[..., unk_idx] = torch.clamp(probs[..., unk_idx] - unk_penalty, min=0.0)
    return probs


def top_p_mask(probs: torch.Tensor, p: float) -> torch.Tensor:
    """TopPSampler: sort descending, drop rank r when (cumsum - prob)[r] > p.  [..., V] bool."""
    sp, idx = torch.sort(probs, dim=-1, descending=True, stable=True)
    drop = (torch.cumsum(sp, dim=-1) - sp) > p
    return torch.zeros_like(probs, dtype=torch.bool).scatter(-1, idx, ~drop)


def top_k_mask(probs: torch.Tensor, k: int) -> torch.Tensor:
    """TopKSampler: the k largest (ties: lowest token id first)."""
    _, idx = torch.sort(probs, dim=-1, descending=True, stable=True)
    keep = torch.zeros_like(probs, dtype=torch.bool)
    return keep.scatter(-1, idx[..., : min(k, probs.shape[-1])], True)


def sample_filter(logits: torch.Tensor, sampler: Tuple[str, float], temperature: float = 1.0, pad_idx: int = 0,
                  eos_idx: int = 3, block_eos: bool = False, unk_idx: int = 1, unk_penalty: float = 0.0) -> torch.Tensor:
    """Kept set of one sampling step ([..., V] bool; masked tokens are never kept)."""
    probs = sampling_probs(logits, temperature, pad_idx, eos_idx, block_eos, unk_idx, unk_penalty)
    unk_dead = unk_penalty != 0.0 and bool((probs[..., unk_idx] <= 0).all())
    if sampler[0] == "top_k":
        # rank the maskable tokens last even when their zero ties with underflowed probabilities
        ranked = probs.clone()
        ranked[..., pad_idx] = -1.0
        if block_eos:
            ranked[..., eos_idx] = -1.0
        if unk_dead:
            ranked[..., unk_idx] = -1.0
        keep = top_k_mask(ranked, int(sampler[1]))
    else:
        keep = top_p_mask(probs, float(sampler[1]))
    keep[..., pad_idx] = False
    if block_eos:
        keep[..., eos_idx] = False
    if unk_dead:
        keep[..., unk_idx] = False
    return keep


def splitmix_word(seed: int, row: int, step: int) -> int:
    """The 64-bit random word of (seed, sentence, step) -- csrc/sampling.hip smp_hash."""
    z = (seed + 0x9E3779B97F4A7C15 * (row * 65536 + step + 1)) & _MASK64
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _MASK64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _MASK64
    return z ^ (z >> 31)


def q40_masses(logits: torch.Tensor, temperature: float = 1.0, unk_idx: int = 1, unk_penalty: float = 0.0):
    """Integer token masses floor(exp(l/T - max) * 2^40) of one row (numpy uint64) and the max; with an UNK penalty
    mass[unk] -= int(penalty * Z) (never below 0), Z = the sum of the untouched masses (csrc/sampling.hip)."""
    import numpy as np

    t = (logits.float() * np.float32(1.0 / temperature)).numpy().astype(np.float32)
    m = t.max()
    e = np.exp2(((t - m) * np.float32(1.4426950408889634)).astype(np.float32)).astype(np.float32)
    w = (e * np.float32(2.0 ** 40)).astype(np.uint64)
    if unk_penalty != 0.0:
        z = int(w.astype(object).sum())
        w[unk_idx] = max(int(w[unk_idx]) - int(float(np.float32(unk_penalty)) * float(z)), 0)
    return w, float(m)


def sample_draw(masses, keep, z: int) -> Tuple[int, float]:
    """Token whose span of the kept integer mass holds floor(z * kept / 2^64), walking the vocabulary
    in the device's order: thread t of 1024 owns the 4-token groups g = i * 1024 + t, i = 0, 1, ...
    Returns (token, distance of the target to the nearest span boundary / kept mass)."""
    import numpy as np

    v = masses.shape[0]
    idx = np.arange(v)
    g = idx // 4
    order = np.lexsort((idx % 4, g // SAMPLE_THREADS, g % SAMPLE_THREADS))   # thread, iteration, lane-element
    w = np.where(keep[order], masses[order], 0).astype(object)
    csum = np.cumsum(w)
    kept = int(csum[-1])
    target = (z * kept) >> 64
    pos = int(np.searchsorted(np.array(csum, dtype=object), target, side="right"))
    lo = int(csum[pos - 1]) if pos else 0
    margin = min(target - lo, int(csum[pos]) - 1 - target) / max(kept, 1)
    return int(order[pos]), margin


@torch.inference_mode()
def sampling_generate(params, cfg: OracleTextDecoderConfig, embeddings: torch.Tensor, prompt: Sequence[int],
                      sampler: Tuple[str, float], seed: int, min_gen_len: int = 1,
                      max_gen_len: Tuple[int, int] = (1, 128), max_seq_len: Optional[int] = None,
                      normalize_scores: bool = True, len_penalty: float = 1.0, temperature: float = 1.0,
                      pad_idx: int = 0, eos_idx: int = 3, row_offset: int = 0, source_len: Optional[int] = None,
                      unk_idx: int = 1, unk_penalty: float = 0.0):
    """SamplingSeq2SeqGenerator, one hypothesis per embedding; sampler = ("top_k", k) | ("top_p", p).
    Returns [(tokens after the prompt, score, step log-probs)] per embedding."""
    model_max = max_seq_len if max_seq_len is not None else cfg.max_seq_len
    plen = len(prompt)
    if source_len is None:
        source_len = cfg.cond_dim
    max_len = min(plen + int(max_gen_len[0] * source_len + max_gen_len[1]), model_max)
    min_len = min(plen + min_gen_len, max_len)
    out = []
    for r, e in enumerate(embeddings):
        seq, cum, steps = list(prompt), 0.0, []
        if plen > 1:
            lp = torch.log_softmax(decoder_logits(params, cfg, e.unsqueeze(0), torch.tensor([seq[:-1]])) / temperature,
                                   dim=-1, dtype=torch.float32)
            cum = float(lp[0, torch.arange(plen - 1), torch.tensor(seq[1:])].sum())
        step_nr = plen
        while True:
            logits = decoder_logits(params, cfg, e.unsqueeze(0), torch.tensor([seq]))[0, -1]
            lprobs = torch.log_softmax(logits.float() / temperature, dim=-1, dtype=torch.float32)
            if step_nr == max_len - 1:
                tok = eos_idx
            else:
                keep = sample_filter(logits, sampler, temperature, pad_idx, eos_idx, step_nr < min_len, unk_idx, unk_penalty)
                masses, _ = q40_masses(logits, temperature, unk_idx, unk_penalty)
                tok, _ = sample_draw(masses, keep.numpy(), splitmix_word(seed, row_offset + r, step_nr))
                if unk_penalty != 0.0 and tok == unk_idx:   # the step score is log(probs[token]) of the penalised probs
                    lprobs = torch.log(sampling_probs(logits, temperature, pad_idx, eos_idx, False, unk_idx, unk_penalty))
            seq.append(tok)
            cum += float(lprobs[tok])
            steps.append(float(lprobs[tok]))
            step_nr += 1
            if tok == eos_idx:
                break
        score = cum / (step_nr - 1) ** len_penalty if normalize_scores else cum
        out.append((seq[plen:], score, steps))
    return out
