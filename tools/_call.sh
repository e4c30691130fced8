python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids | tail -3
timeout 600 python -m pytest tests -m gpu -x -q -k "xsim or rccl or sampling" 2>&1 | tail -2
