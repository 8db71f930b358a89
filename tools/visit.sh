#!/bin/bash
rm -f ffpa_attn_amd/variants/*.so
FFPA_GIT_HEAD=d4e9af7 bash tools/gpu_evidence.sh
