"""Every switchable code path of the device pipeline must give the same bits: the parity suite is re-run in a
subprocess under each combination of the CFR_* switches (derived tables on/off, v1/v2 search kernel, tiny
sub-batches, sparse locate memo).  -m gpu."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

VARIANTS = {
    "search_v1": {"CFR_SEARCH_V1": "1"},
    "no_derived_tables": {"CFR_FTABX_WIDTH": "0", "CFR_TEXT_MODE": "0", "CFR_LOC_MEMO_GB": "0"},
    "text_mode_early": {"CFR_TEXT_MIN_L": "8", "CFR_FTABX_WIDTH": "0"},
    "tiny_subbatches_sparse_memo": {"CFR_SUBBATCH": "37", "CFR_LOC_MEMO_GB": "0.0003", "CFR_TAPER_FLOOR": "0"},
    # no suffix array: the sparse memo is what the locate walks stop at (multi-kernel locate path)
    "sparse_memo_without_text_mode": {"CFR_TEXT_MODE": "0", "CFR_LOC_MEMO_GB": "0.0003", "CFR_SUBBATCH": "41"},
    # the lean image of a 40 Gbp index forced on the small ones: 36-bit packed suffix array, 8-byte K-mer entries, no locate
    # memo (every row is located through the suffix array and the step function)
    "lean_image": {"CFR_FORCE_WIDE": "1", "CFR_FTABX_E8": "1", "CFR_LOC_MEMO_GB": "0"},
    "lean_image_text_mode_early": {"CFR_FORCE_WIDE": "1", "CFR_FTABX_E8": "1", "CFR_FTABX_WIDTH": "8", "CFR_LOC_MEMO_GB": "0", "CFR_TEXT_MIN_L": "8"},
    "no_memo_locate_by_steps": {"CFR_LOC_MEMO_GB": "0"},
    # SDUST of resident reads sub-batch by sub-batch on the dust stream (the bench's with_device_sdust leg and the CLI use the device mask)
    "dust_by_pieces": {"CFR_DUST_PIECES": "1", "CFR_SUBBATCH": "64", "CFR_TAPER_FLOOR": "0"},
    "ftabx_8_byte_entries": {"CFR_FTABX_E8": "1", "CFR_FTABX_WIDTH": "12"},
    # round 6: the table the large indexes get by default - K = 17, 8-byte entries (34-bit keys; 137 GB whatever the index) - on the golden reads
    "ftabx_k17": {"CFR_FTABX_WIDTH": "17", "CFR_FTABX_E8": "1"},
    # ... and the two-launch search (stage 1 without wide text mode + stage 2 over the list: off by default, kept measurable)
    "search_split_many_subbatches_wave_tiles": {"CFR_SEARCH_SPLIT": "1", "CFR_SEARCH_DYN": "2", "CFR_SUBBATCH": "53", "CFR_TAPER_FLOOR": "0"},
    # the sampled rows "do not follow" the step function: no text mode, locate memo by the plain walk
    "step_function_rejected": {"CFR_STEPS_OFF": "1"},
    "two_kernel_post_stage": {"CFR_FUSED_POST": "0"},
    "post_pool_overflow_redo": {"CFR_POOL_CAP": "3", "CFR_SUBBATCH": "50", "CFR_TAPER_FLOOR": "0"},
    "post_pool_growth": {"CFR_POOL_INIT": "3", "CFR_SUBBATCH": "50", "CFR_TAPER_FLOOR": "0"},
    "host_inputs_staged_up_front": {"CFR_STREAM_INPUTS": "0"},
    "fast_load_profile": {"CFR_PROFILE": "fast-load"},
    "tapered_last_subbatch": {"CFR_SUBBATCH": "300", "CFR_TAPER_FLOOR": "5"},
    "unfused_tail": {"CFR_FUSED_TAIL": "0", "CFR_SUBBATCH": "61"},
    "wide_ftabx_one_block_per_cu": {"CFR_FTABX_WIDTH": "12", "CFR_BLOCKS_PER_CU": "1"},
    # the reference's own compressed components in HBM (rank lines, wavelet trees, run-block rank): whole parity file
    "run_block_layout": {"CFR_LAYOUT": "rb"},
    # the n >= 2^32 code path (36-bit packed SA entries, WIDE search kernel) forced on the small indexes
    "wide_tables": {"CFR_FORCE_WIDE": "1"},
    "wide_tables_text_mode_early": {"CFR_FORCE_WIDE": "1", "CFR_TEXT_MIN_L": "8", "CFR_FTABX_WIDTH": "0"},
    # ranges of 5 .. 24 rows continue on the text ("wide text mode"): off, and cut down to 6 rows
    "no_wide_text_mode": {"CFR_WIDE_ROWS": "0"},
    "wide_text_mode_6_rows_early": {"CFR_WIDE_ROWS": "6", "CFR_TEXT_MIN_L": "8", "CFR_FTABX_WIDTH": "0"},
    # the locate memo built by the plain LF walk per row instead of from the text order
    "memo_by_walk": {"CFR_MEMO_WALK": "1"},
    # reads with many located rows folded by one lane through the pool (tail_fold_hash) instead of a team (k_tail_heavy)
    "no_team_tail": {"CFR_TEAM_TAIL": "0"},
    # the post stage on the search's own stream (no overlap with the next sub-batch's search), with several sub-batches
    "post_stage_on_the_search_stream": {"CFR_TAIL_STREAM": "0", "CFR_SUBBATCH": "97", "CFR_TAPER_FLOOR": "0"},
    "post_stage_overlapped_many_subbatches": {"CFR_TAIL_STREAM": "1", "CFR_SUBBATCH": "53", "CFR_TAPER_FLOOR": "0"},
    "post_stage_overlapped_grid_stride": {"CFR_TAIL_STREAM": "1", "CFR_TAIL_BLOCKS": "1", "CFR_SUBBATCH": "1000", "CFR_TAPER_FLOOR": "0"},
    "run_block_layout_plain": {"CFR_LAYOUT": "rb", "CFR_FTABX_WIDTH": "0", "CFR_LOC_MEMO_GB": "0"},
    # hit-list offsets per sub-batch (k_caps + scan in front of every search) instead of one pass over the resident batch
    "hit_offsets_per_sub_batch": {"CFR_CAPS_ONCE": "0", "CFR_SUBBATCH": "43", "CFR_TAPER_FLOOR": "0"},
    # round 5: the common read's post stage by k_post_fast (registers only), k_adjust_tail over the reads it lists
    "post_fast": {"CFR_POST_FAST": "1"},
    "post_fast_many_subbatches": {"CFR_POST_FAST": "1", "CFR_SUBBATCH": "53", "CFR_TAPER_FLOOR": "0", "CFR_POST_FAST_LAST": "1"},
    "post_fast_on_the_search_stream": {"CFR_POST_FAST": "1", "CFR_TAIL_STREAM": "0", "CFR_SUBBATCH": "97"},
    # chains handed out by wave tiles (the default for pairs) forced everywhere / off everywhere / per-lane draws of 8
    "search_wave_tiles": {"CFR_SEARCH_DYN": "2"},
    "search_wave_tiles_of_128_small_subbatches": {"CFR_SEARCH_DYN": "2", "CFR_SEARCH_TILE": "128", "CFR_SUBBATCH": "61"},
    "search_static_hand_out": {"CFR_SEARCH_DYN": "0"},
    "search_lane_draws": {"CFR_SEARCH_DYN": "1"},
    # the searches of odd sub-batches on a second stream (measured: no gain; kept as a switch)
    "two_search_streams": {"CFR_SEARCH_TWO": "1", "CFR_SUBBATCH": "53", "CFR_TAPER_FLOOR": "0"},
    # K-mer entries without the text position of their one row (the form before round 5)
    "ftabx_without_text_positions": {"CFR_FTABX_TEXTPOS": "0"},
}


@pytest.mark.parametrize("name", sorted(VARIANTS))
def test_parity_suite_under_switches(name):
    env = dict(os.environ, CFR_DEBUG_ENV="1")     # the gate that makes the library look at its CFR_* test switches at all
    env.update(VARIANTS[name])
    select = [] if name.startswith("run_block") else ["-k", "tsv or hit_lists or backward_search or degenerate or fresh_index or derived_tables"]
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_parity.py"), "-m", "gpu", "-x", "-q"] + select,
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=ROOT)
    tail = r.stdout.decode()[-1500:]
    assert r.returncode == 0, f"{name} {VARIANTS[name]}:\n{tail}"
    assert " passed" in tail


@pytest.mark.parametrize("name,extra", [("wide_tables", {"CFR_FORCE_WIDE": "1"}), ("no_wide_text_mode", {"CFR_WIDE_ROWS": "0"}),
                                        ("lean_image", {"CFR_FORCE_WIDE": "1", "CFR_FTABX_E8": "1", "CFR_LOC_MEMO_GB": "0"}),
                                        ("no_team_tail", {"CFR_TEAM_TAIL": "0"}), ("team_tail_k1", {"CFR_TEST_K": "1"}), ("team_tail_k5", {"CFR_TEST_K": "5"}),
                                        ("post_stage_overlapped", {"CFR_TAIL_STREAM": "1", "CFR_SUBBATCH": "20000"}),
                                        ("post_stage_never_overlapped", {"CFR_TAIL_STREAM": "0", "CFR_SUBBATCH": "20000"}),
                                        ("post_fast", {"CFR_POST_FAST": "1", "CFR_SUBBATCH": "20000"}), ("post_fast_k5", {"CFR_POST_FAST": "1", "CFR_TEST_K": "5"}),
                                        ("search_wave_tiles", {"CFR_SEARCH_DYN": "2", "CFR_SUBBATCH": "20000"}), ("search_static", {"CFR_SEARCH_DYN": "0"}),
                                        ("search_split", {"CFR_SEARCH_SPLIT": "1", "CFR_SUBBATCH": "20000"})])
def test_many_strain_workload_under_switches(name, extra):
    """The 20-strain workload (ranges of up to 20 rows: wide text mode, hash fold) of tests/test_gpu_scale.py with the 5-byte
    tables forced (the WIDE kernel's wide text mode on a small index) and with wide text mode off."""
    env = dict(os.environ, CFR_DEBUG_ENV="1")
    env.update(extra)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_scale.py"), "-m", "gpu", "-x", "-q", "-k", "strain"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=ROOT)
    tail = r.stdout.decode()[-1500:]
    assert r.returncode == 0, f"{name}:\n{tail}"
    assert " passed" in tail


@pytest.mark.parametrize("name,extra", [("protein_two_arrays", {"CFR_PROT_TWO_ARRAYS": "1"}), ("protein_no_kmer_table", {"CFR_FTABX_WIDTH": "0"}),
                                        ("protein_kmer_table_3", {"CFR_FTABX_WIDTH": "3"}), ("protein_kmer_table_5_two_arrays", {"CFR_FTABX_WIDTH": "5", "CFR_PROT_TWO_ARRAYS": "1"})])
def test_protein_suite_under_switches(name, extra):
    """tests/test_gpu_protein.py with the protein image's other forms: counts and planes in two arrays (what texts of 2^32 symbols
    and more get) instead of one 128-byte record per block, and the derived K-mer table off / at other widths."""
    env = dict(os.environ, CFR_DEBUG_ENV="1")
    env.update(extra)
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_protein.py"), "-m", "gpu", "-x", "-q"],
                       env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, cwd=ROOT)
    tail = r.stdout.decode()[-1500:]
    assert r.returncode == 0, f"{name}:\n{tail}"
    assert " passed" in tail
