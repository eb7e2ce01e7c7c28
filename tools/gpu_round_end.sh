# Final validation of a build (smoke, the whole -m gpu suite, the default bench line), then the round profile.
bash tools/gpu_final.sh && R=r02 bash tools/run_profile.sh
