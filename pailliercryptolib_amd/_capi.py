"""ctypes binding of the C-ABI in include/pgpu.h (libpgpu.so, built in-tree by build.py).

This module is plumbing only: it loads the HIP library and declares argument types.  There is
deliberately no fallback: if the library is missing or no gfx950 device is present, every
compute call raises.
"""
import ctypes
import os
from ctypes import c_char_p, c_double, c_int, c_size_t, c_void_p, POINTER, c_uint64

_HERE = os.path.dirname(os.path.abspath(__file__))
# PGPU_LIB: an alternative build of the library (kernel A/B experiments, tools/build_variant.py)
LIB_PATH = os.environ.get("PGPU_LIB") or os.path.join(_HERE, "libpgpu.so")

# every symbol include/pgpu.h declares (tests check the built library exports all of them)
SYMBOLS = [
    "pgpu_init", "pgpu_shutdown", "pgpu_device_count", "pgpu_is_initialized",
    "pgpu_last_error", "pgpu_device_name",
    "pgpu_modexp", "pgpu_modexp_dev", "pgpu_modmul", "pgpu_modmul_dev",
    "pgpu_pubkey_create", "pgpu_pubkey_destroy", "pgpu_paillier_encrypt", "pgpu_paillier_encrypt_dev",
    "pgpu_privkey_create", "pgpu_privkey_destroy", "pgpu_paillier_decrypt_crt",
    "pgpu_paillier_decrypt_crt_dev",
    "pgpu_dev_alloc", "pgpu_dev_free", "pgpu_copy_h2d", "pgpu_copy_d2h",
    "pgpu_set_fixed_base_window", "pgpu_set_timing", "pgpu_timing_collect",
    "pgpu_kernel_geometry", "pgpu_decrypt_kernel_form", "pgpu_encrypt_kernel_form", "pgpu_modexp_n2_kernel_form", "pgpu_ct_add_kernel_form",
    "pgpu_init_all", "pgpu_pool_size", "pgpu_set_device", "pgpu_get_device", "pgpu_pool_transport",
    "pgpu_set_min_shard", "pgpu_shard_plan", "pgpu_synchronize", "pgpu_set_secret_exponent_policy", "pgpu_get_secret_exponent_policy",
    "pgpu_batch_create", "pgpu_batch_upload", "pgpu_batch_download", "pgpu_batch_destroy", "pgpu_batch_count",
    "pgpu_batch_words", "pgpu_batch_is_montgomery", "pgpu_batch_encrypt", "pgpu_batch_decrypt_crt",
    "pgpu_batch_ct_add", "pgpu_batch_ct_add_plain", "pgpu_batch_ct_mul",
    "pgpu_rccl_note", "pgpu_replication_stats", "pgpu_debug_corrupt_next_replica",
    "pgpu_set_fixed_base_budget", "pgpu_fixed_base_stats", "pgpu_pubkey_fixed_base_info",
    "pgpu_batch_row_limbs", "pgpu_set_batch_lane", "pgpu_batch_lane", "pgpu_batch_download_async", "pgpu_ticket_wait", "pgpu_batch_download_strided",
    "pgpu_set_table_gather_policy", "pgpu_get_table_gather_policy",
    "pgpu_build_features", "pgpu_batch_is_current", "pgpu_batch_lanes", "pgpu_timing_collect_ex", "pgpu_decrypt_kernel_form_ex",
    "pgpu_encrypt_kernel_form_ex", "pgpu_host_alloc", "pgpu_host_free", "pgpu_host_wait",
    "pgpu_timing_collect_trace",
]
FEATURE_4096_SPLIT = 1

_lib = None


class PgpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"pgpu error {code}: {msg}")
        self.code = code


def lib():
    """Load libpgpu.so (once).  Raises if the extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    u64p = POINTER(c_uint64)
    L.pgpu_init.argtypes = [c_int]; L.pgpu_init.restype = c_int
    L.pgpu_shutdown.argtypes = []; L.pgpu_shutdown.restype = None
    L.pgpu_device_count.argtypes = []; L.pgpu_device_count.restype = c_int
    L.pgpu_is_initialized.argtypes = []; L.pgpu_is_initialized.restype = c_int
    L.pgpu_last_error.argtypes = []; L.pgpu_last_error.restype = c_char_p
    L.pgpu_device_name.argtypes = []; L.pgpu_device_name.restype = c_char_p
    L.pgpu_modexp.argtypes = [c_void_p, c_size_t, c_void_p, c_size_t, c_int, c_int, c_void_p, c_int,
                              c_void_p, c_size_t]
    L.pgpu_modexp.restype = c_int
    L.pgpu_modexp_dev.argtypes = [c_void_p, c_size_t, c_void_p, c_size_t, c_int, c_int, c_void_p,
                                  c_int, c_void_p, c_size_t, c_void_p]
    L.pgpu_modexp_dev.restype = c_int
    L.pgpu_modmul.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_void_p, c_size_t]
    L.pgpu_modmul.restype = c_int
    L.pgpu_modmul_dev.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p, c_int, c_void_p, c_size_t,
                                  c_void_p]
    L.pgpu_modmul_dev.restype = c_int
    L.pgpu_pubkey_create.argtypes = [c_void_p, c_int, c_void_p, POINTER(c_void_p)]
    L.pgpu_pubkey_create.restype = c_int
    L.pgpu_pubkey_destroy.argtypes = [c_void_p]; L.pgpu_pubkey_destroy.restype = None
    L.pgpu_paillier_encrypt.argtypes = [c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_size_t, c_int,
                                        c_int, c_void_p, c_size_t]
    L.pgpu_paillier_encrypt.restype = c_int
    L.pgpu_paillier_encrypt_dev.argtypes = [c_void_p, c_void_p, c_size_t, c_int, c_void_p, c_size_t,
                                            c_int, c_int, c_void_p, c_size_t, c_void_p]
    L.pgpu_paillier_encrypt_dev.restype = c_int
    L.pgpu_privkey_create.argtypes = [c_void_p, c_void_p, c_int, POINTER(c_void_p)]
    L.pgpu_privkey_create.restype = c_int
    L.pgpu_privkey_destroy.argtypes = [c_void_p]; L.pgpu_privkey_destroy.restype = None
    L.pgpu_paillier_decrypt_crt.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t]
    L.pgpu_paillier_decrypt_crt.restype = c_int
    L.pgpu_paillier_decrypt_crt_dev.argtypes = [c_void_p, c_void_p, c_void_p, c_size_t, c_void_p]
    L.pgpu_paillier_decrypt_crt_dev.restype = c_int
    L.pgpu_dev_alloc.argtypes = [c_size_t, POINTER(c_void_p)]; L.pgpu_dev_alloc.restype = c_int
    L.pgpu_dev_free.argtypes = [c_void_p]; L.pgpu_dev_free.restype = None
    L.pgpu_copy_h2d.argtypes = [c_void_p, c_void_p, c_size_t]; L.pgpu_copy_h2d.restype = c_int
    L.pgpu_copy_d2h.argtypes = [c_void_p, c_void_p, c_size_t]; L.pgpu_copy_d2h.restype = c_int
    L.pgpu_set_fixed_base_window.argtypes = [c_int]; L.pgpu_set_fixed_base_window.restype = c_int
    L.pgpu_set_timing.argtypes = [c_int]; L.pgpu_set_timing.restype = c_int
    L.pgpu_timing_collect.argtypes = [c_void_p, c_void_p, c_int]; L.pgpu_timing_collect.restype = c_int
    L.pgpu_kernel_geometry.argtypes = [c_int, c_int, c_size_t, POINTER(c_int), POINTER(c_int)]
    L.pgpu_kernel_geometry.restype = c_int
    L.pgpu_decrypt_kernel_form.argtypes = [c_void_p, c_size_t, POINTER(c_int), POINTER(c_int), POINTER(c_int)]
    L.pgpu_decrypt_kernel_form.restype = c_int
    L.pgpu_encrypt_kernel_form.argtypes = [c_void_p, c_int, c_size_t, POINTER(c_int), POINTER(c_int), POINTER(c_int)]
    L.pgpu_encrypt_kernel_form.restype = c_int
    L.pgpu_modexp_n2_kernel_form.argtypes = [c_void_p, c_size_t, POINTER(c_int), POINTER(c_int), POINTER(c_int)]
    L.pgpu_modexp_n2_kernel_form.restype = c_int
    L.pgpu_ct_add_kernel_form.argtypes = [c_void_p, c_size_t, POINTER(c_int), POINTER(c_int), POINTER(c_int)]
    L.pgpu_ct_add_kernel_form.restype = c_int
    L.pgpu_init_all.argtypes = [c_int]; L.pgpu_init_all.restype = c_int
    L.pgpu_pool_size.argtypes = []; L.pgpu_pool_size.restype = c_int
    L.pgpu_set_device.argtypes = [c_int]; L.pgpu_set_device.restype = c_int
    L.pgpu_get_device.argtypes = []; L.pgpu_get_device.restype = c_int
    L.pgpu_pool_transport.argtypes = []; L.pgpu_pool_transport.restype = c_char_p
    L.pgpu_set_min_shard.argtypes = [c_size_t]; L.pgpu_set_min_shard.restype = c_int
    L.pgpu_synchronize.argtypes = []; L.pgpu_synchronize.restype = c_int
    L.pgpu_shard_plan.argtypes = [c_size_t, c_int, POINTER(c_int), POINTER(c_size_t)]
    L.pgpu_shard_plan.restype = c_int
    L.pgpu_set_secret_exponent_policy.argtypes = [c_int]; L.pgpu_set_secret_exponent_policy.restype = c_int
    L.pgpu_get_secret_exponent_policy.argtypes = []; L.pgpu_get_secret_exponent_policy.restype = c_int
    L.pgpu_batch_create.argtypes = [c_size_t, c_int, POINTER(c_void_p)]; L.pgpu_batch_create.restype = c_int
    L.pgpu_batch_upload.argtypes = [c_void_p, c_size_t, c_int, c_size_t, POINTER(c_void_p)]
    L.pgpu_batch_upload.restype = c_int
    L.pgpu_batch_download.argtypes = [c_void_p, c_void_p]; L.pgpu_batch_download.restype = c_int
    L.pgpu_batch_download_async.argtypes = [c_void_p, c_void_p, POINTER(c_void_p)]; L.pgpu_batch_download_async.restype = c_int
    L.pgpu_ticket_wait.argtypes = [c_void_p]; L.pgpu_ticket_wait.restype = c_int
    L.pgpu_batch_download_strided.argtypes = [c_void_p, c_void_p, c_size_t]; L.pgpu_batch_download_strided.restype = c_int
    L.pgpu_batch_destroy.argtypes = [c_void_p]; L.pgpu_batch_destroy.restype = None
    L.pgpu_batch_count.argtypes = [c_void_p]; L.pgpu_batch_count.restype = c_size_t
    L.pgpu_batch_words.argtypes = [c_void_p]; L.pgpu_batch_words.restype = c_int
    L.pgpu_batch_is_montgomery.argtypes = [c_void_p]; L.pgpu_batch_is_montgomery.restype = c_int
    L.pgpu_batch_encrypt.argtypes = [c_void_p, c_void_p, c_void_p, c_int, POINTER(c_void_p)]
    L.pgpu_batch_encrypt.restype = c_int
    L.pgpu_batch_decrypt_crt.argtypes = [c_void_p, c_void_p, POINTER(c_void_p)]
    L.pgpu_batch_decrypt_crt.restype = c_int
    L.pgpu_batch_ct_add.argtypes = [c_void_p, c_void_p, c_void_p, POINTER(c_void_p)]
    L.pgpu_batch_ct_add.restype = c_int
    L.pgpu_batch_ct_add_plain.argtypes = [c_void_p, c_void_p, c_void_p, POINTER(c_void_p)]
    L.pgpu_batch_ct_add_plain.restype = c_int
    L.pgpu_batch_ct_mul.argtypes = [c_void_p, c_void_p, c_void_p, c_int, POINTER(c_void_p)]
    L.pgpu_batch_ct_mul.restype = c_int
    L.pgpu_rccl_note.argtypes = []; L.pgpu_rccl_note.restype = c_char_p
    L.pgpu_replication_stats.argtypes = [POINTER(c_uint64), POINTER(c_uint64)]; L.pgpu_replication_stats.restype = c_int
    L.pgpu_debug_corrupt_next_replica.argtypes = [c_int]; L.pgpu_debug_corrupt_next_replica.restype = c_int
    L.pgpu_set_fixed_base_budget.argtypes = [c_size_t, c_size_t]; L.pgpu_set_fixed_base_budget.restype = c_int
    L.pgpu_fixed_base_stats.argtypes = [c_int, POINTER(c_size_t), POINTER(c_uint64)]; L.pgpu_fixed_base_stats.restype = c_int
    L.pgpu_pubkey_fixed_base_info.argtypes = [c_void_p, c_int, POINTER(c_int), POINTER(c_size_t), POINTER(c_double)]
    L.pgpu_pubkey_fixed_base_info.restype = c_int
    L.pgpu_batch_row_limbs.argtypes = [c_void_p]; L.pgpu_batch_row_limbs.restype = c_int
    L.pgpu_set_batch_lane.argtypes = [c_int]; L.pgpu_set_batch_lane.restype = c_int
    L.pgpu_batch_lane.argtypes = [c_void_p]; L.pgpu_batch_lane.restype = c_int
    L.pgpu_set_table_gather_policy.argtypes = [c_int]; L.pgpu_set_table_gather_policy.restype = c_int
    L.pgpu_get_table_gather_policy.argtypes = []; L.pgpu_get_table_gather_policy.restype = c_int
    L.pgpu_build_features.argtypes = []; L.pgpu_build_features.restype = c_int
    L.pgpu_batch_lanes.argtypes = []; L.pgpu_batch_lanes.restype = c_int
    L.pgpu_host_alloc.argtypes = [c_size_t, POINTER(c_void_p)]; L.pgpu_host_alloc.restype = c_int
    L.pgpu_host_free.argtypes = [c_void_p]; L.pgpu_host_free.restype = None
    L.pgpu_host_wait.argtypes = [c_void_p]; L.pgpu_host_wait.restype = c_int
    L.pgpu_timing_collect_trace.argtypes = [c_void_p, c_void_p, c_void_p, c_void_p, c_void_p, c_int]
    L.pgpu_timing_collect_trace.restype = c_int
    L.pgpu_timing_collect_ex.argtypes = [c_void_p, c_void_p, c_void_p, c_int]; L.pgpu_timing_collect_ex.restype = c_int
    L.pgpu_decrypt_kernel_form_ex.argtypes = [c_void_p, c_size_t, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]
    L.pgpu_decrypt_kernel_form_ex.restype = c_int
    L.pgpu_encrypt_kernel_form_ex.argtypes = [c_void_p, c_int, c_size_t, c_int, POINTER(c_int), POINTER(c_int), POINTER(c_int)]
    L.pgpu_encrypt_kernel_form_ex.restype = c_int
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise PgpuError(rc, lib().pgpu_last_error().decode(errors="replace"))
