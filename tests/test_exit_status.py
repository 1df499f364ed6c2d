"""Exit status (and stdout) of the product CLI against the oracle CLI on unreadable / malformed inputs.

The reference `.expect()`s its FASTX inputs (src/lib.rs:144,149,221,234; src/cluster.rs:28,39): a missing, empty,
non-FASTX or truncated input is a PANIC — exit 101 — while the DB file is opened with `?` (src/lib.rs:208-210,214,218)
and `count` propagates everything with `?` (src/lib.rs:381,385): Err out of main — exit 1.  A DB file shorter than four
bytes panics on the slice `&buffer[0..4]` (src/lib.rs:214), an old version panics by hand (:215-217).
Cases that end before any scan run here without a GPU; the ones that print rows first are marked gpu.
"""
import gzip
import os
import subprocess

import pytest

import oracle
from smafa_amd import _lib

FQ = b"@r1\nACGTACGT\n+\nIIIIIIII\n@r2\nACGTACGA\n+\nIIIIIIII\n"
FA = b">a\nACGTACGT\n>b\nACGTACGA\n>c\nTTTTACGA\n"


def product(*args):
    return subprocess.run([_lib.CLI_PATH, *args], capture_output=True)


def reference(*args):
    return subprocess.run([oracle.CLI, *args], capture_output=True)


def same(args, want_code):
    p, r = product(*args), reference(*args)
    assert r.returncode == want_code, (args, r.returncode, r.stderr)
    assert p.returncode == r.returncode, (args, p.returncode, r.returncode, p.stderr, r.stderr)
    assert p.stdout == r.stdout, (args, p.stdout, r.stdout)
    assert p.stderr.strip() != b""  # says why
    return p


@pytest.fixture()
def files(tmp_path):
    d = {}

    def put(name, data):
        path = str(tmp_path / name)
        with open(path, "wb") as f:
            f.write(data)
        d[name] = path
        return path

    put("ok.fa", FA)
    put("empty.fa", b"")
    put("garbage.fa", b"this is not a sequence file\n")
    put("blank_first.fa", b"\n>a\nACGT\n")
    put("trunc.fq", FQ + b"@r3\nACGTAC")          # record cut inside the sequence line
    put("trunc_plus.fq", FQ + b"@r3\nACGTACGT\n")  # no '+' line
    put("qual_short.fq", FQ + b"@r3\nACGTACGT\n+\nIII\n")
    put("mixed.fa", FA + b"@r3\nACGTACGT\n+\nIIIIIIII\n")  # FASTQ record inside a FASTA file
    put("short.db", b"\x02\x00")                   # fewer than 4 bytes: slice panic
    put("v1.db", bytes.fromhex("0102") + b"\x00" * 32)
    put("body.db", bytes.fromhex("0205 01c810"))   # five windows announced, one present: postcard Err
    with gzip.open(str(tmp_path / "trunc.fq.gz"), "wb") as g:
        g.write(FQ + b"@r3\nACGTAC")
    d["trunc.fq.gz"] = str(tmp_path / "trunc.fq.gz")
    d["missing"] = str(tmp_path / "does" / "not" / "exist")
    d["out"] = str(tmp_path / "out.db")
    r = reference("makedb", "-i", d["ok.fa"], "-d", str(tmp_path / "ok.db"))
    assert r.returncode == 0
    d["ok.db"] = str(tmp_path / "ok.db")
    return d


@pytest.mark.parametrize("name", ["missing", "empty.fa", "garbage.fa", "blank_first.fa", "trunc.fq", "trunc_plus.fq",
                                  "qual_short.fq", "mixed.fa", "trunc.fq.gz"])
def test_makedb_input_failures_are_panics(files, name):
    same(["makedb", "-i", files[name], "-d", files["out"]], 101)  # src/lib.rs:144,149


def test_makedb_output_failure_is_an_err(files):
    same(["makedb", "-i", files["ok.fa"], "-d", os.path.join(files["missing"], "x.db")], 1)  # File::create(..)?, :161


@pytest.mark.parametrize("name", ["missing", "empty.fa", "garbage.fa", "blank_first.fa"])
def test_query_file_failures_are_panics(files, name):
    same(["query", "-d", files["ok.db"], "-q", files[name]], 101)  # src/lib.rs:221


@pytest.mark.parametrize("db,code", [("missing", 1), ("empty.fa", 101), ("short.db", 101), ("v1.db", 101), ("body.db", 1)])
def test_query_db_failures(files, db, code):
    # File::open(db)? -> 1; a file below four bytes panics on the slice (an empty file too); old version panics;
    # a truncated body is a postcard Err -> 1.  The DB is looked at before the query file: a bad query file changes nothing.
    same(["query", "-d", files[db], "-q", files["ok.fa"]], code)
    same(["query", "-d", files[db], "-q", files["missing"]], code)


@pytest.mark.parametrize("name", ["missing", "empty.fa", "garbage.fa", "blank_first.fa"])
def test_cluster_input_failures_are_panics(files, name):
    same(["cluster", "-i", files[name], "-d", "2"], 101)  # src/cluster.rs:28


@pytest.mark.parametrize("name", ["missing", "empty.fa", "garbage.fa", "trunc.fq", "qual_short.fq"])
def test_count_failures_are_errs(files, name):
    same(["count", "-i", files[name]], 1)  # src/lib.rs:381,385: `?`


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["trunc.fq", "trunc_plus.fq", "qual_short.fq", "mixed.fa", "trunc.fq.gz"])
def test_bad_record_after_good_ones(files, name):
    """record.expect(..) in the loops (src/lib.rs:234; src/cluster.rs:39): the records in front of the bad one are answered
    (same stdout), then the panic: exit 101"""
    p = same(["query", "-d", files["ok.db"], "-q", files[name], "--max-divergence", "8"], 101)
    assert p.stdout != b""
    p = same(["cluster", "-i", files[name], "-d", "2"], 101)
    assert p.stdout != b""


def test_no_subcommand_prints_help_and_succeeds():
    """src/main.rs:52-56: `smafa` alone prints the help text (stdout) and returns Ok(())"""
    p = product()
    assert p.returncode == 0 and b"Usage: smafa" in p.stdout and b"makedb" in p.stdout and b"cluster" in p.stdout
    for flag in ("-v", "-q", "--verbose", "--quiet"):  # the top-level flags alone are still "no subcommand"
        p = product(flag)
        assert p.returncode == 0 and b"Usage: smafa" in p.stdout


@pytest.mark.parametrize("flag", ["-v", "-q", "--verbose", "--quiet"])
def test_top_level_verbosity_flags_in_front_of_the_subcommand(files, flag):
    """src/main.rs:67-68 declares -v/--verbose and -q/--quiet on the top-level command too: `smafa -q makedb ..` parses"""
    p = product(flag, "makedb", "-i", files["ok.fa"], "-d", files["out"])
    r = reference("makedb", "-i", files["ok.fa"], "-d", files["out"] + ".ref")
    assert p.returncode == 0 and r.returncode == 0, p.stderr
    assert open(files["out"], "rb").read() == open(files["out"] + ".ref", "rb").read()
    p = product(flag, "count", "-i", files["ok.fa"])
    assert p.returncode == 0 and p.stdout == reference("count", "-i", files["ok.fa"]).stdout
