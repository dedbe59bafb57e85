#!/bin/bash
# Kernel ms (u8, u16) of the scan matcher on the 10,000-pair bench workload under environment settings:
#   tools/bnb_env_ab.sh "NAME=VALUE ..." "NAME=VALUE ..." ...      (an empty string: the defaults)
cd "$(dirname "$0")/.."
for e in "$@"; do echo "[$e]: $(env $e timeout -k 5 100 python3 tools/bnb_quick.py 2>/dev/null | tr '\n' ' ')"; done
