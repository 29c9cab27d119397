export TMPDIR=/tmp
python tools/measure/dropin_prof.py flat > gpurun_out/dropin_prof_flat.txt 2>&1 && python tools/measure/dropin_prof.py adam > gpurun_out/dropin_prof_adam.txt 2>&1; head -3 gpurun_out/dropin_prof_flat.txt; head -3 gpurun_out/dropin_prof_adam.txt
