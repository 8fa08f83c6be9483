"""tests/test_gpu_e2e_distill.py with the LPIPS module forced to bf16 operands (the r05 arithmetic): the A/B of the r06 IEEE-half default."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import pytest
import sparsefusion_amd.lpips as L
_init = L.LPIPS.__init__
def init(self, *a, **k):
    _init(self, *a, **k)
    self.operand, self.grad_scale = None, 1.0
L.LPIPS.__init__ = init
sys.exit(pytest.main([os.path.join(ROOT, "tests", "test_gpu_e2e_distill.py"), "-m", "gpu", "-q", "-s"]))
