( timeout -k 10 500 python tests/tools/fuzz_parity.py 400 7 | tail -3 ) > gpurun_out/r2n_fuzz.log 2>&1
( PPP_FUZZ_TINY=1 timeout -k 10 300 python tests/tools/fuzz_parity.py 300 8 | tail -3 ) >> gpurun_out/r2n_fuzz.log 2>&1
( PPP_FUZZ_ODD=1 timeout -k 10 300 python tests/tools/fuzz_parity.py 150 9 | tail -3 ) >> gpurun_out/r2n_fuzz.log 2>&1
( PPP_FUZZ_PRE=1 timeout -k 10 300 python tests/tools/fuzz_parity.py 60 10 | tail -3 ) >> gpurun_out/r2n_fuzz.log 2>&1
( PPP_FUZZ_BIG=1 timeout -k 10 400 python tests/tools/fuzz_parity.py 12 11 | tail -3 ) >> gpurun_out/r2n_fuzz.log 2>&1
cat gpurun_out/r2n_fuzz.log
