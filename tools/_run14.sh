for s in 0 1 2 4 8 256 257 258 260 0; do PITA_EGNN_STAGGER=$s python tools/time_sampler.py; done > gpurun_out/r3_stagger.log 2>&1
