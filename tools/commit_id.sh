#!/bin/bash
# Build container: the id profile summaries carry as `collected_at_commit` -- short hash of HEAD, "+dirty" when the
# tree differs from it (pass it to the tools/gpu_*.sh scripts; the GPU box has no .git).
cd "$(dirname "$0")/.." && echo "$(git rev-parse --short HEAD)$(git diff --quiet HEAD -- . || echo +dirty)"
