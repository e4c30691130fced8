#!/bin/bash
# ONE command for a person who HAS the released SONAR files: run the reference's own real-checkpoint goldens
# (tests/test_gpu_reference_goldens.py = /root/reference/tests/integration_tests/test_text_sonar.py:46-161,
# test_sonar_speech_encoder.py:56-78, test_sonar_speech_pipeline_models.py:28-60, restated) on the MI355X engine
# and save the measured deltas under profiles/.
#
#   SONAR_CHECKPOINT_DIR=/path/to/files bash tools/run_reference_goldens.sh [tag]
#
# Files looked for in $SONAR_CHECKPOINT_DIR (names = the last URL component of the reference's asset cards,
# sonar_amd/cards.py):  sonar_text_encoder.pt  sonar_text_decoder.pt  spenc.eng.pt  sentencepiece.source.256000.model
# A test whose files are missing is SKIPPED (and listed as such in the summary); nothing is downloaded.
set -u
TAG=${1:-goldens}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT"
if [ -z "${SONAR_CHECKPOINT_DIR:-}" ] || [ ! -d "$SONAR_CHECKPOINT_DIR" ]; then
  echo "set SONAR_CHECKPOINT_DIR to the directory that holds the released checkpoints" >&2
  exit 2
fi
mkdir -p profiles
LOG=profiles/${TAG}_reference_goldens_full.log
OUT=profiles/${TAG}_reference_goldens.txt
python -m pytest tests/test_gpu_reference_goldens.py -m gpu -v -s -rs 2>&1 | tee "$LOG"
{
  echo "# reference goldens on the MI355X engine -- $(date -u +%Y-%m-%dT%H:%M:%SZ), SONAR_CHECKPOINT_DIR=$SONAR_CHECKPOINT_DIR"
  echo "# files present:"
  for f in sonar_text_encoder.pt sonar_text_decoder.pt spenc.eng.pt sentencepiece.source.256000.model; do
    if [ -e "$SONAR_CHECKPOINT_DIR/$f" ]; then echo "#   $f ($(stat -c %s "$SONAR_CHECKPOINT_DIR/$f") bytes)"; else echo "#   $f MISSING"; fi
  done
  echo "# TOLERANCES: the reference asserts its fp32 CPU run to 1e-4 (cosine matrix, logits, speech embeddings: rtol 1e-4 /"
  echo "#   atol 1e-5 of torch.testing.assert_close) -- this engine multiplies fp16 operands with fp32 accumulation, so the"
  echo "#   restated tests hold embeddings to 1e-3 on cosine quantities (BASELINE north_star) and fp16-model logits to 5e-2"
  echo "#   absolute; a PASSED line below is a pass at THOSE bounds, the measured deltas are printed so the reference's own"
  echo "#   bounds can be read off directly.  Token ids and translated strings are compared exactly."
  echo "# measured deltas (printed by the tests):"
  grep -E "cosine matrix|max \|diff\||1 - cos|dot products|^\[\[|tokens|translat" "$LOG"
  echo "# outcome per test:"
  grep -E "PASSED|FAILED|SKIPPED|ERROR" "$LOG" | sed 's/^/  /'
  tail -1 "$LOG"
} > "$OUT"
echo "summary written to $OUT"
