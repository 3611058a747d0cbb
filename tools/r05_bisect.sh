cd "$GRAFT_REPO_ROOT"
for e in "SAU_AMD_NO_CHAIN_INLINE=1" "SAU_AMD_NO_EARLY_CHAINS=1" "SAU_AMD_NO_LEAN=1" "SAU_AMD_NO_MIX_FEW=1" "SAU_AMD_NO_SEQ=1"; do
echo "== $e"; env SAU_AMD_TUNE=1 $e python tests/tools/debug_sweep_seed.py 3501608 2>&1 | grep "drop-in default:"
done
SAU_AMD_DEBUG=1 python tests/tools/debug_sweep_seed.py 3501608 2>&1 | grep -v "step \|amdgpu" | head -40
