import os, sys, time
sys.path.insert(0, ".")
import torch
from oracle import text_encoder as O
cfg = O.OracleTextEncoderConfig()
params = O.make_synthetic_params(cfg, seed=1234)
ids, _ = O.synthetic_batch(8, 128, 128, cfg.vocab_size, seed=0)
print("cpu_count", os.cpu_count(), flush=True)
for th in (16, 32, 64, 128):
    torch.set_num_threads(th)
    O.text_encoder_forward(params, cfg, ids[:1], None)
    t0 = time.time(); O.text_encoder_forward(params, cfg, ids, None); dt = time.time() - t0
    print(th, "threads:", 8 / dt, "sent/s", flush=True)
