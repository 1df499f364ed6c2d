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
for f in common alphabet fastx dbfile select drivers; do
  g++ -O1 -g -std=c++17 -fPIC -pthread -fsanitize=address,undefined -fno-omit-frame-pointer -c host/$f.cpp -o "$OUT/$f.o"
done
g++ -O1 -g -std=c++17 -pthread -fsanitize=address,undefined -c host/main.cpp -o "$OUT/main.o"
g++ -fsanitize=address,undefined -pthread -o "$OUT/smafa_asan" "$OUT"/*.o ../build/engine.o -L/opt/rocm/lib -lamdhip64 -lz -lpthread -Wl,-rpath,/opt/rocm/lib
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
synth.write_fasta(out + "/big.fna", synth.subjects(1_200_000, 60, 0, seed=2, n_frac=0.001), 0)
synth.write_fasta(out + "/small.fna", synth.subjects(300, 70, 0, seed=3, n_frac=0.01), 0)
run("makedb", "-i", out + "/big.fna", "-d", out + "/big.db"); run("query", "-d", out + "/big.db", "-q", out + "/small.fna")
run("cluster", "-i", out + "/big.fna", "-d", "3"); run("count", "-i", out + "/big.fna")
run("makedb", "-i", out + "/small.fna", "-d", out + "/small.db")
db = open(out + "/small.db", "rb").read(); fa = open(out + "/small.fna", "rb").read()
rng = random.Random(5)
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
print("sanitizer findings:", findings)
sys.exit(1 if findings else 0)
PY
