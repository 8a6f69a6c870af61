"""GPU end-to-end: the overlap command's file output, the torch-side merge helper on one GPU, and
size-independent properties at BASELINE.json's full config-2 size."""
import io
import os

import numpy as np
import pytest
import torch

import checker as ck
import golden_utils as gu
from oracle import overlap_oracle as oo   # row helpers only; the oracle itself runs in the checker process
from phasm_amd import cli, synth
from phasm_amd.io import gfa
from phasm_amd.overlapper import ExactOverlapper

pytestmark = pytest.mark.gpu


def test_cli_overlap_writes_reference_lines(tmp_path):
    """H line, all S lines, then E lines byte-identical to the reference CLI's formatting
    (assembler.py:30,38,46-48) for the golden rows."""
    name, seqs, m, want = gu.ladder_case("ladder_small")
    cfg = synth.SynthConfig(n_reads=80, read_len=2000, genome_len=10_000, ploidy=2, snp=0.005, seed=11)
    reads = synth.generate_reads(cfg)
    fa = tmp_path / "reads.fasta"
    synth.write_fasta(str(fa), reads, width=70)           # multi-line records
    out = tmp_path / "out.gfa"
    assert cli.main(["overlap", str(fa), "-l", str(m), "-o", str(out)]) == 0
    out_py = tmp_path / "out_py.gfa"
    assert cli.main(["overlap", str(fa), "-l", str(m), "-o", str(out_py), "--python-ingest"]) == 0
    assert out.read_bytes() == out_py.read_bytes()   # native ingest == Python ingest, byte for byte
    import os
    os.environ["PHASM_FASTA_RANGES"] = "5"            # the parallel record scan cut into ranges (a 160 kB file is one range otherwise)
    os.environ["PHASM_POISON_HOST"] = "1"             # ... into stores that start as 0xA5: the ingest writes every word itself
    try:
        out_r = tmp_path / "out_ranges.gfa"
        assert cli.main(["overlap", str(fa), "-l", str(m), "-o", str(out_r)]) == 0
    finally:
        del os.environ["PHASM_FASTA_RANGES"]
        del os.environ["PHASM_POISON_HOST"]
    assert out_r.read_bytes() == out.read_bytes()
    lines = out.read_text().splitlines(keepends=True)
    assert lines[0] == "H\tVN:z:2.0\n"
    s_lines = [l for l in lines if l.startswith("S\t")]
    e_lines = [l for l in lines if l.startswith("E\t")]
    assert lines[1:1 + len(s_lines)] == [gfa.gfa_line("S", n, len(s), "*") for n, s in reads]
    assert len(lines) == 1 + len(s_lines) + len(e_lines)
    ids = [n for n, _ in synth.oriented(reads)]
    want_lines = sorted(gfa.gfa_line("E", "*", ids[a], ids[b], s, e, bs, be, "*") for a, b, s, e, bs, be in want.tolist())
    assert sorted(e_lines) == want_lines


def test_native_and_python_edge_writers_agree(tmp_path):
    _, seqs, m, want = gu.ladder_case("ladder_varlen")
    ov = ExactOverlapper()
    ids = []
    for i, s in enumerate(seqs):
        ids.append("read %d/x%s" % (i // 2, "+-"[i % 2]))
        ov.add_sequence(ids[-1], s)
    res = ov.overlaps_result(m)
    path = tmp_path / "edges.gfa"
    with open(path, "w") as f:
        f.write("H\tVN:z:2.0\n")
        assert res.write_gfa_edges(f) == len(want)
    py = io.StringIO()
    gfa.write_edges(py, res.rows(), ids)
    res.free()
    ov.close()
    assert path.read_text() == "H\tVN:z:2.0\n" + py.getvalue()


def test_merge_helper_single_gpu():
    from phasm_amd.dist import rows_tensor_to_struct, sharded_overlaps
    _, seqs, m, want = gu.ladder_case("ladder_varlen")
    ov = ExactOverlapper(device=0)
    for i, s in enumerate(seqs):
        ov.add_sequence("r%d" % i, s)
    t = sharded_overlaps(ov, m, device=torch.device("cuda", 0))
    assert t.is_cuda and t.dtype == torch.int32 and t.shape[1] == 6
    assert np.array_equal(oo.sort_rows(oo.struct_to_rows(rows_tensor_to_struct(t))), want)
    ov.close()


_RCCL_CHILD = r"""
import os, socket, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import golden_utils as gu
from oracle import overlap_oracle as oo
from phasm_amd.dist import CandidateExchange, rows_tensor_to_struct, sharded_overlaps
from phasm_amd.overlapper import ExactOverlapper
with socket.socket() as sk:          # a port nobody holds, instead of a fixed one
    sk.bind(("127.0.0.1", 0))
    port = sk.getsockname()[1]
os.environ["MASTER_ADDR"] = "127.0.0.1"
os.environ["MASTER_PORT"] = str(port)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
try:
    _, seqs, m, want = gu.ladder_case("ladder_cfg2_mini")
    ov = ExactOverlapper(device=0)
    for i, s in enumerate(seqs):
        ov.add_sequence("r%%d" %% i, s)
    t = sharded_overlaps(ov, m, device=dev)
    whole = ov.overlaps_array(m)
    assert np.array_equal(rows_tensor_to_struct(t), whole)
    assert np.array_equal(oo.sort_rows(oo.struct_to_rows(whole)), want)
    # the repeated-step form: first step sizes the slot (a stricter min_length: fewer candidates), the second
    # outgrows it (slot grows in the middle of the step), the third and fourth write straight into the slot
    ex = CandidateExchange(ov, device=dev, slack=1.0)
    r0 = ex.rows(4 * m)
    assert len(r0) < len(whole)
    r0.free()
    for _ in range(3):
        r = ex.rows(m)
        assert np.array_equal(r.rows(), whole)
        r.free()
    assert ex.n_collectives == 2 + 2 + 1 + 1
    # the pipelined step: own rows expanded and copied home while the (async) all-gather of the candidates is in flight;
    # a smaller and a larger problem in turn (the slot must grow in the middle of a pipelined step as well)
    for mm in (m, 4 * m, m, m):
        merged, own = ex.rows_home(mm)
        got = oo.sort_rows(oo.struct_to_rows(own.rows_view()))
        own.free()
        ref = oo.sort_rows(oo.struct_to_rows(ov.overlaps_array(mm)))
        assert np.array_equal(got, ref), (mm, len(got), len(ref))
        assert merged.shape[1] == 4 and merged.is_cuda
    # the sharded upload in parts through the same code path (one rank: the plain upload is taken)
    from phasm_amd.dist import ReadExchange
    assert ReadExchange(ov, device=dev).upload(parts=3) is False
    ov.close()
finally:
    dist.destroy_process_group()
print("RCCL OK")
"""


def test_rccl_collectives_one_rank_group():
    """The N>1 merge goes through RCCL (`nccl` backend).  A one-rank process group on this GPU runs the
    very same calls (count all-gather, padded all_gather_into_tensor of int32[n,4], expansion) -- in a child
    process: a process group brings RCCL's proxy threads and pinned buffers with it, and the rest of the suite
    should not share an address space with them."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = _RCCL_CHILD % {"root": root, "tests": os.path.join(root, "tests")}
    rc, stdout, stderr = ck.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=500)
    assert rc == 0 and "RCCL OK" in stdout, stdout[-2000:] + stderr[-3000:]


@pytest.mark.skipif(os.environ.get("PHASM_SKIP_FULL") == "1", reason="full-size run disabled")
def test_full_size_cfg2_properties():
    """50k x 15 kb (100k oriented reads): too big for the CPU oracle in seconds -- the rows are compared with the
    generator-derived expected multiset (exact), and on top with what must
    hold at any size: every row is a true match on the original strings (sampled), A rows are unique
    per (a,b) and reach the end of a, B rows cover the whole of b, the multiset is closed under the
    strand mirror, and a sampled set of reads agrees with the CPU oracle run on
    just those reads' neighbourhoods."""
    cfg = synth.CONFIGS["cfg2"]
    reads = synth.generate_reads(cfg)
    oriented = synth.oriented(reads)
    ov = ExactOverlapper(device=0)
    for n, s in oriented:
        ov.add_sequence(n, s)
    res = ov.overlaps_result(1000)
    arr = res.rows()
    st = ov.stats()
    # the consumer on the same rows, still in HBM: layout stage 1 at full size against the numpy restatement
    # (cheap enough at 7 M rows) -- every edge, bit for bit; plus the sharded form of the overlap call
    from oracle import layout_oracle as lo
    from phasm_amd import layout
    g = layout.build_assembly_graph(ov, res, min_read_length=0, min_overlap_length=2000)
    res.free()
    want_g = ck.layout_vectorised(oo.struct_to_rows(arr), ov.lengths(), min_overlap_length=2000)
    e = g.edges
    got_e = np.stack([e["u"], e["v"], e["weight"], e["overlap_len"]], 1).astype(np.int64)
    got_e = got_e[np.lexsort((got_e[:, 1], got_e[:, 0]))]
    assert np.array_equal(got_e, want_g["edges"]) and len(got_e) > 5_000_000
    assert g.contained.tolist() == want_g["contained"].tolist()
    sharded = np.concatenate([ov.overlaps_shard_array(1000, k, 3) for k in range(3)])
    assert np.array_equal(oo.sort_rows(oo.struct_to_rows(sharded)), oo.sort_rows(oo.struct_to_rows(arr)))
    del sharded
    ov.close()
    # THE WHOLE MULTISET, exactly: what the generator's truth says the reference must return for these reads
    # (synth.expected_rows, pinned to the reference on all ladder goldens by tests/test_synth_truth.py)
    import rowsig
    want_all = synth.expected_rows(cfg, 1000)
    rowsig.assert_same_multiset(oo.struct_to_rows(arr), want_all, "cfg2 at full size")
    assert np.array_equal(oo.sort_rows(oo.struct_to_rows(arr)), oo.sort_rows(want_all))
    del want_all
    lens = np.array([len(s) for _, s in oriented], dtype=np.int64)
    a, b = arr["a_idx"].astype(np.int64), arr["b_idx"].astype(np.int64)
    s, e, bs, be = (arr[k].astype(np.int64) for k in ("astart", "aend", "bstart", "bend"))
    assert len(arr) == st["n_rows"] > 6_000_000
    assert (a != b).all() and (bs == 0).all() and (be >= 1000).all() and (e - s == be).all()
    is_a = e == lens[a]
    is_b = be == lens[b]
    assert (is_a | is_b).all()
    # sampled string check on the original bytes
    rng = np.random.default_rng(0)
    for i in rng.integers(0, len(arr), size=3000):
        assert oriented[a[i]][1][s[i]:e[i]] == oriented[b[i]][1][:be[i]]
    # A rows: one per ordered pair
    fam_a = arr[is_a & ~is_b]
    key = fam_a["a_idx"].astype(np.int64) << 32 | fam_a["b_idx"].astype(np.int64)
    assert len(np.unique(key)) == len(key)
    # strand-mirror closure of the whole multiset
    rows = oo.struct_to_rows(arr)
    both = rows[is_a & is_b]
    uniq, cnt = np.unique(both, axis=0, return_counts=True)
    assert (cnt % 2 == 0).all()
    fa_ = np.concatenate([rows[is_a & ~is_b], uniq.repeat(cnt // 2, axis=0)])
    fb_ = np.concatenate([rows[is_b & ~is_a], uniq.repeat(cnt // 2, axis=0)])
    ma = np.stack([fa_[:, 1] ^ 1, fa_[:, 0] ^ 1, lens[fa_[:, 1]] - fa_[:, 5], lens[fa_[:, 1]], fa_[:, 4], fa_[:, 5]], axis=1)
    mb = np.stack([fb_[:, 0] ^ 1, fb_[:, 1] ^ 1, lens[fb_[:, 0]] - fb_[:, 3], lens[fb_[:, 0]] - fb_[:, 2], fb_[:, 4], fb_[:, 5]], axis=1)
    assert np.array_equal(oo.sort_rows(np.concatenate([ma, mb])), oo.sort_rows(rows))
    # oracle on a closed neighbourhood: rows among the first 400 oriented reads must equal the
    # oracle run on those 400 reads alone (rows only depend on the two reads involved)
    sub = [s_ for _, s_ in oriented[:400]]
    want = ck.oracle_overlaps(sub, 1000)
    got = oo.sort_rows(rows[(rows[:, 0] < 400) & (rows[:, 1] < 400)])
    ck.assert_same_rows(got, want, sub, 1000, "cfg2 full size, first 400 oriented reads")


def test_one_hip_runtime_in_the_process():
    """torch bundles its own libamdhip64.so; the library binds to whichever copy is mapped first
    (phasm_amd/_lib.py preloads torch's).  Two HIP runtimes in one address space would each own a device context
    and their own staging threads -- this process must hold exactly one."""
    ov = ExactOverlapper(device=0)
    ov.add_sequence("a", "ACGTACGTAGGCTAGCTAGGATCGATCGATTAGC")
    ov.add_sequence("b", "GATCGATTAGCAAAAACCCCCGGGGGTTTTTACG")
    ov.overlaps_array(5)
    ov.close()
    torch.zeros(4, device="cuda").sum().item()
    paths = set()
    with open("/proc/self/maps") as f:
        for line in f:
            if "libamdhip64" in line:
                paths.add(os.path.realpath(line.split()[-1]))
    assert len(paths) == 1, sorted(paths)


_TWO_RANK_CHILD = r"""
import os, sys
import numpy as np
import torch
import torch.distributed as dist
sys.path.insert(0, %(root)r); sys.path.insert(0, %(tests)r)
import golden_utils as gu
from oracle import overlap_oracle as oo
from phasm_amd.dist import CandidateExchange, rows_tensor_to_struct, _result_to_tensor
from phasm_amd.overlapper import ExactOverlapper
os.environ["PHASM_INDEX"] = "wide"
dist.init_process_group("gloo")
rank, ws = dist.get_rank(), dist.get_world_size()
try:
    _, seqs, m, want = gu.ladder_case("cfg3_1k")
    ov = ExactOverlapper(device=0)
    for i, s in enumerate(seqs):
        ov.add_sequence("r%%d" %% i, s)
    from phasm_amd.dist import ReadExchange
    rx = ReadExchange(ov)
    assert rx.upload(parts=3) is True and rx.n_collectives == 3     # each rank uploaded its share in three parts, the rest came over the wire
    assert rx.upload() is True                                      # (default: one part for a piece this small)
    ex = CandidateExchange(ov, device=torch.device("cpu"))          # gloo carries the collectives, one GPU runs both ranks
    for step in range(2):
        res = ex.rows(m)
        rows = oo.sort_rows(oo.struct_to_rows(res.rows()))
        res.free()
        assert np.array_equal(rows, want), (rank, step, len(rows), len(want))
        assert ov.stats()["wide_index"] == 1
    # the pipelined step: every rank brings home the rows of ITS shard; together they are the golden rows
    for step in range(2):
        merged, own = ex.rows_home(m)
        mine = oo.struct_to_rows(own.rows_view()).copy()
        own.free()
        parts = [None] * ws
        dist.all_gather_object(parts, mine)
        rows = oo.sort_rows(np.concatenate(parts))
        assert np.array_equal(rows, want), (rank, step, len(rows), len(want))
        assert 0 < len(mine) < len(want)
    assert ex.index.index is not None and ex.index.index["n_slices"] == ws      # the sliced index was exchanged ...
    assert ex.index.n_collectives == 2                                           # ... once: the second step reused it
    ov.close()
finally:
    dist.destroy_process_group()
print("RANK %%d OK" %% rank)
"""


@pytest.mark.parametrize("nproc", [2, 4])
def test_two_ranks_on_one_gpu_exchange_the_sliced_index(nproc, tmp_path):
    """The N > 1 step end to end with two (and four) ranks sharing this GPU (gloo carries the collectives; RCCL refuses two
    ranks on one device): rank g builds sub-table g of the wide index, the chunks are all-gathered, every rank's
    shard probes the gathered index, candidates are exchanged and expanded -- both ranks end with the golden rows."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / "two_ranks.py"
    script.write_text(_TWO_RANK_CHILD % {"root": root, "tests": os.path.join(root, "tests")})
    rc, stdout, stderr = ck.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % nproc,
                                 "--master-addr", "127.0.0.1", "--master-port", str(29641 + nproc), str(script)],
                                capture_output=True, text=True, timeout=500)
    assert rc == 0 and all("RANK %d OK" % r in stdout for r in range(nproc)), stdout[-2000:] + stderr[-3000:]
