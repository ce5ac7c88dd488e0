#!/bin/bash
for e in "A=1" "VOX_ATTN_F32=1" "VOX_NO_SKINNY=1" "VOX_ATTN_F32=1 VOX_NO_SKINNY=1"; do
echo "--- $e"
env $e timeout 600 python -m pytest tests/test_gpu_model.py -m gpu -q --timeout 300 -k "encode_audio" 2>&1 | tail -4
done
