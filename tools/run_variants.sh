for f in rope_s3d_amd/csrc/librope_hip_var_*.so; do echo "== $f"; ROPE_HIP_LIB=$PWD/$f timeout -k 10 120 python tools/profile_phases.py 2>&1 | grep -E "^full  |no loss pass" ; done
