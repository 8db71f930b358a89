#!/bin/bash
rm -f ffpa_attn_amd/variants/*.so
FFPA_GIT_HEAD=37e79a9 bash tools/gpu_evidence.sh
