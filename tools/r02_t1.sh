#!/bin/bash
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "p2p_exchange_on_one_gpu" 2>&1 | tail -5
