import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """The C-ABI library is a build product (git-ignored): build it in-tree when a fresh checkout runs
    the tests before __graft_entry__.build() (hipcc cross-compiles gfx950 without a GPU).  A failing
    build is left for the tests to report -- importing the package raises loudly without the .so."""
    so = os.path.join(ROOT, "robust-dynrf_amd", "librodynrf.so")
    if os.environ.get("RDRF_LIB") or os.path.exists(so):
        return
    import shutil
    import subprocess
    if shutil.which("make") and (shutil.which("hipcc") or os.path.exists("/opt/rocm/bin/hipcc")):
        subprocess.call(["make", "-C", os.path.join(ROOT, "robust-dynrf_amd", "csrc"), "-j8"],
                        stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)


# ------------------------------------------------------------------------------------------------------------------
# Collection order of the GPU suite (the driver runs `pytest tests -x -q -m gpu`: a failure must never hide a test that
# speaks for parity with the REFERENCE).  Tier 0: comparisons with fixtures the reference itself generated
# (tests/golden/*.npz / *.th: forward 10-tuples, compositor outputs, autograd gradients, the pass structure, complete
# train.reconstruction() iterations, reference-written checkpoints).  Tier 1: comparisons with the oracle (the pinned CPU
# restatement) or another independent implementation (torch.optim.Adam, F.interpolate, brute-force formulas) at sizes the
# fixtures do not cover.  Tier 2: self-consistency and property tests (bit-identity of two code paths, batched vs per-pass,
# additivity, determinism, edge cases, error paths), the subprocess-driven bench / multi-rank runs last.
# Inside a tier the file / definition order is kept.
# ------------------------------------------------------------------------------------------------------------------
TIER0_REFERENCE = {
    "test_golden_forward", "test_raygen_golden", "test_golden_gradients", "test_raygen_gradients_golden",
    "test_golden_ray_gradients", "test_golden_function_vectors", "test_golden_function_gradients",
    "test_sample_ray_wrappers", "test_reference_written_checkpoint_evaluates_like_the_reference",
    "test_induce_flow_golden", "test_render_single_3d_point_golden", "test_dense_l1_matches_reference_fixture",
    "test_tvloss_single_tensor_golden", "test_tv_family_golden", "test_l1_and_ortho_regularisers_golden",
    "test_pass_structure_matches_reference_fixture", "test_trainer_step_matches_reference_trainer_iteration",
}
TIER1_ORACLE = {
    "test_oracle_forward_balloon_shapes", "test_render_frame_matches_oracle_pipeline", "test_oracle_gradients_midsize",
    "test_oracle_gradients_midsize_contract", "test_oracle_gradients_benchmark_grids",
    "test_full_batch_gradients_without_ray_exclusion", "test_z_vals_gradient_matches_oracle",
    "test_sorted_scatter_passes_the_same_parity_tests", "test_function_gradients_vs_oracle",
    "test_induce_flow_oracle_bench_shape", "test_distortion_loss_vs_bruteforce", "test_trainer_step_gradient_matches_oracle_step",
    "test_loss_terms_match_the_torch_expressions", "test_frame_depth_loss_matches_reference_loop",
    "test_adam_step_matches_torch_adam", "test_flat_adam_trains_like_torch_adam_on_the_fields",
    "test_dense_l1_balloon_grid_vs_einsum", "test_upsample_kernel_matches_interpolate", "test_pruned_branches_match_autograd",
    "test_fused_grad_accumulation_matches_autograd", "test_tv_accumulate_grad_matches_autograd_path",
    "test_tv_family_foreign_callable_matches", "test_generate_rays_uv_and_view_shift", "test_selftest_mfma_layer",
}
TIER3_SUBPROCESS = {
    "test_two_rank_step_with_exact_statistics_matches_single_process", "test_bench_two_ranks_functional",
    "test_bench_rccl_call_sequence_on_one_gpu", "test_bench_refuses_world_size_mismatch",
    "test_deterministic_build_is_bit_reproducible_and_matches_the_atomic_build",
    "test_render_chunks_on_hip_streams_is_bit_identical_in_the_deterministic_build",
}


def parity_tier(item):
    name = item.originalname if getattr(item, "originalname", None) else item.name.split("[")[0]
    if name in TIER0_REFERENCE:
        return 0
    if name in TIER1_ORACLE:
        return 1
    return 3 if name in TIER3_SUBPROCESS else 2


def pytest_collection_modifyitems(session, config, items):
    gpu = [i for i, it in enumerate(items) if it.get_closest_marker("gpu") is not None]
    if not gpu:
        return
    ordered = sorted((items[i] for i in gpu), key=parity_tier)   # stable: file / definition order inside a tier
    for slot, it in zip(gpu, ordered):
        items[slot] = it
