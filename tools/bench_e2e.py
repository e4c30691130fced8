"""End-to-end probe of TextToEmbeddingModelPipeline.predict on one MI355X: strings in, embeddings
out, tokenisation + host input path + PCIe included (the judged bench.py times the engine with ids
already resident in HBM).  Trains a synthetic 32k-piece unigram SentencePiece model in /tmp (there is
no real NLLB model in this environment) and compares the C++ host path with the per-sentence one."""
import os
import random
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def make_corpus(n, seed=0):
    rnd = random.Random(seed)
    syll = ["ka", "lo", "mi", "ten", "sur", "pa", "ri", "vo", "da", "ne", "shi", "bu", "tra", "el", "on", "qu", "ix", "za"]
    words = ["".join(rnd.choice(syll) for _ in range(rnd.randint(1, 4))) for _ in range(30000)]
    cum, acc = [], 0.0
    for i in range(len(words)):   # Zipf-like
        acc += 1.0 / (i + 1) ** 0.9
        cum.append(acc)
    return [" ".join(rnd.choices(words, cum_weights=cum, k=rnd.randint(80, 120))) for _ in range(n)]


def main():
    import sentencepiece as spm

    from sonar_amd.inference_pipelines.text import TextToEmbeddingModelPipeline
    from sonar_amd.text_encoder import SonarTextTransformerEncoderModel, get_text_encoder_config
    from sonar_amd.tokenizer import NllbTokenizer
    from tools.synth import text_encoder_state_dict

    n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
    texts = make_corpus(n)
    with open("/tmp/e2e_corpus.txt", "w") as fh:
        fh.write("\n".join(texts[:20000]))
    t0 = time.time()
    spm.SentencePieceTrainer.train(input="/tmp/e2e_corpus.txt", model_prefix="/tmp/e2e", vocab_size=32000,
                                   model_type="unigram", hard_vocab_limit=False, minloglevel=2)
    print(f"trained synthetic SPM model in {time.time() - t0:.1f}s", flush=True)
    tok = NllbTokenizer("/tmp/e2e.model")
    enc = tok.create_encoder(lang="eng_Latn")
    t0 = time.time()
    ids = enc.encode_batch(texts[:8192])
    dt = time.time() - t0
    ntok = sum(map(len, ids))
    print(f"tokenise only (SentencePiece batch encode, all threads): {8192 / dt:.0f} sentences/s, "
          f"{ntok / 8192:.1f} tokens/sentence", flush=True)

    dev = torch.device("cuda:0")
    model = SonarTextTransformerEncoderModel(get_text_encoder_config("basic"), text_encoder_state_dict(dev), device=dev)
    pipe = TextToEmbeddingModelPipeline(model, tok, device=dev)
    for mode in ("native", "python"):
        pipe.host_input = mode
        m = n if mode == "native" else min(n, 8192)
        pipe.predict(texts[:2048], source_lang="eng_Latn", batch_size=1024)   # warm-up
        torch.cuda.synchronize()
        t0 = time.time()
        out = pipe.predict(texts[:m], source_lang="eng_Latn", batch_size=1024)
        torch.cuda.synchronize()
        dt = time.time() - t0
        print(f"predict() end to end, host path {mode:6s}: {m / dt:8.0f} sentences/s ({m} sentences, batch 1024, "
              f"{dt * 1e3:.0f} ms); out {tuple(out.shape)} finite={bool(torch.isfinite(out).all())}", flush=True)


if __name__ == "__main__":
    main()
