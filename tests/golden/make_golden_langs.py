"""Generate tests/golden/nllb_language_codes.json -- the 202 NLLB language codes in fairseq dictionary
order, taken from HuggingFace `transformers.models.nllb.tokenization_nllb.FAIRSEQ_LANGUAGE_CODES` (an
independent source of the order the SONAR tokenizer card lists, text_sonar_basic_encoder.yaml:14-216).
The language token of code i is id 256001 + i (SentencePiece pieces shifted by one behind <pad>), e.g.
eng_Latn -> 256047, the id the reference's notebook shows.

Run in the build container:  python tests/golden/make_golden_langs.py
"""
import json
import os

from transformers.models.nllb.tokenization_nllb import FAIRSEQ_LANGUAGE_CODES

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "nllb_language_codes.json")

if __name__ == "__main__":
    json.dump(list(FAIRSEQ_LANGUAGE_CODES), open(OUT, "w"))
    print("wrote", OUT, len(FAIRSEQ_LANGUAGE_CODES))
