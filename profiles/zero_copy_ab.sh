for m in 0 1 2 3; do
  echo "== NRPHY_DL_SLOT_ZERO_COPY=$m"
  NRPHY_DL_SLOT_ZERO_COPY=$m timeout -k 10 300 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "dl_slot" 2>&1 | tail -1
  NRPHY_DL_SLOT_ZERO_COPY=$m timeout -k 10 200 python3 -c "
import sys; sys.path.insert(0, 'profiles'); sys.path.insert(0, 'tests')
import shim_latency as s
s.dl_slot_pipeline(depths=(1, 4, 8), check=True)
" 2>&1 | grep "slot pipeline" | sed 's/; PCIe per slot.*host time/; host time/'
done
