for dbg in 0 1 4 5 8 13; do echo "VF_PWG_DBG=$dbg"; VF_PWG_DBG=$dbg timeout -k 10 120 python scripts/bench_pwgrad.py 64 2>/dev/null | grep " dW " ; done
