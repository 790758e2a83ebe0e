timeout -k 10 200 python tools/stream_scale.py
BF_GEMM_STREAM=0 timeout -k 10 200 python tools/stream_scale.py
