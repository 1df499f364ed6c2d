#!/bin/bash
# AddressSanitizer + UBSan over the HOST side (FASTX reader, DB file reader/writer, drivers, selection, CLI) on a CPU-only
# machine: the host objects are rebuilt with -fsanitize=address,undefined and linked with the ordinary engine.o; every
# command runs up to the point where it needs a device.  GPU sanitizers are not available on the pool.
#   tools/asan_host.sh            -> builds /tmp/asan/smafa_asan, runs it over tests/golden, a 1.2M-record FASTA
#                                    (the parallel ingest paths) and ~1000 truncated / mutated DB and FASTA files
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${OUT:-/tmp/asan}
mkdir -p "$OUT"
make -C "$ROOT/smafa_amd/csrc" -j8 >/dev/null
cd "$ROOT/smafa_amd/csrc"
for f in common group qsession inflate_other alphabet fastx dbfile select packed layout drivers; do
  g++ -O1 -g -std=c++17 -fPIC -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -c host/$f.cpp -o "$OUT/$f.o"
done
g++ -O1 -g -std=c++17 -pthread -fsanitize=address,undefined -c host/main.cpp -o "$OUT/main.o"
g++ -fsanitize=address,undefined -pthread -o "$OUT/smafa_asan" "$OUT"/*.o ../build/engine.o -L/opt/rocm/lib -lamdhip64 -lz -lpthread -ldl -Wl,-rpath,/opt/rocm/lib
g++ -O1 -g -std=c++17 -pthread -fsanitize=address,undefined -c "$ROOT/tools/asan_parts.cpp" -o "$OUT/parts_main.oo"
g++ -fsanitize=address,undefined -pthread -o "$OUT/parts_asan" "$OUT/parts_main.oo" $(ls "$OUT"/*.o | grep -v main.o) ../build/engine.o -L/opt/rocm/lib -lamdhip64 -lz -lpthread -ldl -Wl,-rpath,/opt/rocm/lib
cd "$ROOT"
export ASAN_OPTIONS=detect_leaks=0
python3 - "$OUT" <<'PY'
import os, random, subprocess, sys
out = sys.argv[1]
A = os.path.join(out, "smafa_asan")
sys.path.insert(0, os.getcwd())
from smafa_amd import synth
findings = 0
def run(*cmd):
    global findings
    r = subprocess.run([A, *cmd], capture_output=True)
    if b"AddressSanitizer" in r.stderr or b"runtime error" in r.stderr or r.returncode < 0:
        findings += 1
        print("FINDING", cmd, r.returncode, r.stderr[:800].decode(errors="replace"))
    return r
G = "tests/golden"
for f in sorted(os.listdir(G)):
    p = os.path.join(G, f)
    if f.endswith("smafadb"):
        run("query", "-d", p, "-q", os.path.join(G, "random_3_2.fna"))
    else:
        run("makedb", "-i", p, "-d", out + "/g.db"); run("count", "-i", p); run("cluster", "-i", p, "-d", "2")
        run("query", "-d", out + "/g.db", "-q", p)
rng = random.Random(5)
synth.write_fasta(out + "/big.fna", synth.subjects(1_200_000, 60, 0, seed=2, n_frac=0.001), 0)
synth.write_fasta(out + "/small.fna", synth.subjects(300, 70, 0, seed=3, n_frac=0.01), 0)
run("makedb", "-i", out + "/big.fna", "-d", out + "/big.db"); run("query", "-d", out + "/big.db", "-q", out + "/small.fna")
run("cluster", "-i", out + "/big.fna", "-d", "3"); run("count", "-i", out + "/big.fna")
run("makedb", "-i", out + "/small.fna", "-d", out + "/small.db")
# the packed store file written on the host, then read back (query decodes it up to the point where it needs a device)
run("makedb", "-i", out + "/big.fna", "-d", out + "/big.packed", "--packed", "--no-gpu")
run("makedb", "-i", out + "/small.fna", "-d", out + "/small.packed", "--packed", "--no-gpu")
run("query", "-d", out + "/small.packed", "-q", out + "/small.fna")
pk = open(out + "/small.packed", "rb").read()
for _ in range(150):
    d = bytearray(pk)
    for _ in range(rng.randint(1, 5)): d[rng.randrange(len(d))] = rng.randrange(256)
    open(out + "/m.packed", "wb").write(bytes(d[: rng.randrange(8, len(d))] if rng.random() < 0.3 else d))
    run("query", "-d", out + "/m.packed", "-q", out + "/small.fna")
# the threaded FASTQ / gzip loaders (>= 32 MB after decompression), whole and truncated
import gzip
rows = synth.subjects(560_000, 60, 0, seed=4, n_frac=0.001)
letters = b"ACGTN"
fq = b"".join(b"@r%d\n" % i + bytes(letters[c] for c in r) + b"\n+\n" + (b"@" if i % 4 == 0 else b"I") * 60 + b"\n" for i, r in enumerate(rows[:340_000]))
open(out + "/big.fq", "wb").write(fq); open(out + "/big.fq.gz", "wb").write(gzip.compress(fq, 1))
open(out + "/cut.fq", "wb").write(fq[: len(fq) - 37]); open(out + "/cut.fq.gz", "wb").write(gzip.compress(fq, 1)[:-4000])
for f in ("big.fq", "big.fq.gz", "cut.fq", "cut.fq.gz"):
    run("makedb", "-i", out + "/" + f, "-d", out + "/fq.db"); run("count", "-i", out + "/" + f)
# bzip2 / xz input (decoded by the system's libbz2 / liblzma through dlopen): whole, truncated and mutated streams
import bz2, lzma
for name, blob in (("s.bz2", bz2.compress(fa_small := open(out + "/small.fna", "rb").read())), ("s.xz", lzma.compress(fa_small))):
    open(out + "/" + name, "wb").write(blob)
    run("makedb", "-i", out + "/" + name, "-d", out + "/c.db"); run("count", "-i", out + "/" + name)
    for i in range(40):
        d = bytearray(blob)
        for _ in range(rng.randint(1, 3)): d[rng.randrange(4, len(d))] = rng.randrange(256)
        open(out + "/m_" + name, "wb").write(bytes(d[: rng.randrange(6, len(d))] if i % 2 else d))
        run("makedb", "-i", out + "/m_" + name, "-d", out + "/c.db"); run("count", "-i", out + "/m_" + name)
# a packed store header over garbage: the loader must refuse it before anything is indexed
import struct
hdr = b"\x03\x02SMAFA\x00" + struct.pack("<4I11Q", 0, 60, 2, 2, 1000, 4, 1, 4096, 8192, 12288, 16384, 20480, 24576, 28672, 28672 + 4 * 2 * 2 * 256 * 4)
for cut in (len(hdr), 4096, 20000, 40000):
    open(out + "/p.db", "wb").write((hdr + bytes(60000))[:cut]); run("query", "-d", out + "/p.db", "-q", out + "/small.fna")
for _ in range(100):
    d = bytearray(hdr + bytes(range(256)) * 200)
    for _ in range(rng.randint(1, 6)): d[rng.randrange(8, 8 + 120)] = rng.randrange(256)
    open(out + "/p.db", "wb").write(bytes(d)); run("query", "-d", out + "/p.db", "-q", out + "/small.fna")
db = open(out + "/small.db", "rb").read(); fa = open(out + "/small.fna", "rb").read()
cases = [db[:n] for n in list(range(60)) + list(range(60, len(db), 97)) + [len(db) - 1]]
for _ in range(300):
    d = bytearray(db)
    for _ in range(rng.randint(1, 4)): d[rng.randrange(len(d))] = rng.randrange(256)
    cases.append(bytes(d))
for c in cases:
    open(out + "/m.db", "wb").write(c); run("query", "-d", out + "/m.db", "-q", out + "/small.fna")
for i in range(200):
    d = bytearray(fa)
    for _ in range(rng.randint(1, 4)): d[rng.randrange(len(d))] = rng.choice(b">\n\r@+ACGTNx-\x00\xff")
    open(out + "/m.fna", "wb").write(bytes(d[: rng.randrange(1, len(d))] if i % 3 == 0 else d))
    run("makedb", "-i", out + "/m.fna", "-d", out + "/m2.db"); run("count", "-i", out + "/m.fna")
    run("cluster", "-i", out + "/m.fna", "-d", "2"); run("query", "-d", out + "/small.db", "-q", out + "/m.fna")
# the per-part loader of the one-process-per-GPU query path (smafa_fastx_load_part): every part of whole, truncated and mutated
# FASTA / FASTQ files, for several part counts; usable parts must add up to the whole file's records
P = os.path.join(out, "parts_asan")
part_files = [out + "/small.fna", out + "/big.fq", out + "/cut.fq", out + "/big.fq.gz", out + "/big.fna"]
for i in range(60):
    d = bytearray(fa)
    for _ in range(rng.randint(1, 4)): d[rng.randrange(len(d))] = rng.choice(b">\n\r@+ACGTNx-\x00\xff")
    path = out + "/pm%d.fna" % i
    open(path, "wb").write(bytes(d[: rng.randrange(1, len(d))] if i % 3 == 0 else d))
    part_files.append(path)
r = subprocess.run([P, *part_files], capture_output=True)
if b"AddressSanitizer" in r.stderr or b"runtime error" in r.stderr or r.returncode != 0:
    findings += 1
    print("FINDING parts", r.returncode, r.stdout[-600:].decode(errors="replace"), r.stderr[:800].decode(errors="replace"))
print("sanitizer findings:", findings)
sys.exit(1 if findings else 0)
PY
