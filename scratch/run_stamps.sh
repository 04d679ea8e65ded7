cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -Wno-pass-failed -mllvm -amdgpu-mfma-vgpr-form -DMOBROB_STAMPS -o scratch/libmobrob_ppo_stamps.so mobrob_amd/csrc/engine.hip 2>&1 | grep -E "error" | head
python scratch/run_stamps.py 2>&1 | tail -26
