#!/bin/bash
# counters of the default bench's kernels (GPU box): three separate passes (SQ set, FETCH_SIZE, WRITE_SIZE)
bash tools/prof_pmc.sh "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" --no-secondary 2>&1 | tee gpurun_out/pmc_sq.txt
bash tools/prof_pmc.sh "FETCH_SIZE" --no-secondary 2>&1 | tee gpurun_out/pmc_fetch.txt
bash tools/prof_pmc.sh "WRITE_SIZE" --no-secondary 2>&1 | tee gpurun_out/pmc_write.txt
