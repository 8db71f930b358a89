#!/bin/bash
rm -f ffpa_attn_amd/variants/*.so
FFPA_GIT_HEAD=b8c9850 bash tools/gpu_evidence.sh
