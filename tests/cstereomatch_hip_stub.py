"""The reference-side binding of INTEGRATION.md section 2, verbatim (only the library path is
resolved relative to this repository), so that the documented stub is itself under test."""
import ctypes, torch
import os
_lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                "stereo-matching_amd", "stereomatch", "libstereomatch_hip.so"))
_vp, _i, _f, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_size_t
_lib.stm_last_error.restype = ctypes.c_char_p
_lib.stm_ssd_workspace_bytes.restype = _sz
_lib.stm_sgm_workspace_bytes.restype = _sz
_lib.stm_ssd_f32.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _sz, _i, _vp]
_lib.stm_birchfield_f32.argtypes = [_vp, _vp, _vp, _i, _i, _i, _i, _vp]
_lib.stm_sgm_f32.argtypes = [_vp, _vp, _vp, _i, _i, _i, _f, _f, _i, _vp, _sz, _i, _vp]
_lib.stm_wta_f32.argtypes = [_vp, _vp, _i, _i, _i, _i, _vp]
_lib.stm_dp_f32.argtypes = [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _vp]

def _ck(rc):
    if rc: raise RuntimeError(_lib.stm_last_error().decode())
def _s(t): return torch.cuda.current_stream(t.device).cuda_stream
def _ws(n, dev): return torch.empty(max(n, 16), dtype=torch.uint8, device=dev)

class CostOps:
    @staticmethod
    def compute_ssd(left, right, cost_volume, kernel_size):
        H, W = left.shape; D = cost_volume.shape[2]
        ws = _ws(_lib.stm_ssd_workspace_bytes(H, W, D, kernel_size, 1), left.device)
        _ck(_lib.stm_ssd_f32(left.data_ptr(), right.data_ptr(), cost_volume.data_ptr(), H, W, D,
                             kernel_size, 0, ws.data_ptr(), ws.numel(), 1, _s(left)))
    @staticmethod
    def compute_birchfield(left, right, cost_volume, kernel_size):   # kernel_size ignored, as upstream
        H, W = left.shape
        _ck(_lib.stm_birchfield_f32(left.data_ptr(), right.data_ptr(), cost_volume.data_ptr(),
                                    H, W, cost_volume.shape[2], 1, _s(left)))

class AggregationOps:
    @staticmethod
    def run_semiglobal(cost_volume, left_image, penalty1, penalty2, output_cost_volume):
        H, W, D = cost_volume.shape
        n = _lib.stm_sgm_workspace_bytes(H, W, D, 1, 0)
        ws = _ws(n, cost_volume.device)
        # STM_SGM_ACCUMULATE (=1) keeps upstream's "+= into the zero-filled tensor" contract
        # (aggregation.py:51); pass 0 and drop the zero-fill to save one fill and one read.
        _ck(_lib.stm_sgm_f32(cost_volume.data_ptr(), left_image.data_ptr(),
                             output_cost_volume.data_ptr(), H, W, D, penalty1, penalty2, 1,
                             ws.data_ptr(), n, 1, _s(cost_volume)))

class DisparityReduceOps:
    @staticmethod
    def run_winners_take_all(cost_volume, disparity_image):
        H, W, D = cost_volume.shape
        _ck(_lib.stm_wta_f32(cost_volume.data_ptr(), disparity_image.data_ptr(), H, W, D, 1,
                             _s(cost_volume)))
    @staticmethod
    def run_dynamic_programming(cost_volume, path_volume, row_final_costs, disparity_image):
        H, W, D = cost_volume.shape
        _ck(_lib.stm_dp_f32(cost_volume.data_ptr(), path_volume.data_ptr(),
                            row_final_costs.data_ptr(), disparity_image.data_ptr(), H, W, D, 0, 1,
                            _s(cost_volume)))
