PPP_WIN_DEBUG=1 python tools/stamps.py cfg5_10m_s1024 > gpurun_out/stamps_cfg5.txt 2>&1
PPP_WIN_DEBUG=1 python tools/stamps.py cfg2_1m_s256 > gpurun_out/stamps_cfg2.txt 2>&1
cat gpurun_out/stamps_cfg5.txt gpurun_out/stamps_cfg2.txt
