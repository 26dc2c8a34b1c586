bash tools/ab.sh "DS_NO_RESFUSE_X3=1:" "DS_RESFUSE_X3_MIN_NB=2:" "DS_RESFUSE_X3_MIN_NB=4:" "libdiffusynth_hip.so" 2>&1 | tee gpurun_out/r05e/ab2.txt
