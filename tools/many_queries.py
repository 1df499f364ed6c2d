import os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
from smafa_amd import synth, _lib
for n in (1000, 1_000_000):
    subj = synth.subjects(n, 60, 0, seed=2); qry, _, _ = synth.queries(subj, 1_000_000, 0, seed=3, max_subs=6)
    synth.write_fasta("/tmp/ms.fna", subj, 0); synth.write_fasta("/tmp/mq.fna", qry, 0)
    subprocess.run([_lib.CLI_PATH, "makedb", "--quiet", "-i", "/tmp/ms.fna", "-d", "/tmp/ms.db"], check=True)
    for flags in (["--max-divergence", "3"], [], ["--max-num-hits", "3"]):
        t = time.time()
        r = subprocess.run([_lib.CLI_PATH, "query", "-v", "-d", "/tmp/ms.db", "-q", "/tmp/mq.fna", *flags], stdout=open("/tmp/mo.tsv", "wb"), stderr=subprocess.PIPE)
        dt = time.time() - t
        dbg = [l.replace("[DEBUG smafa] ", "") for l in r.stderr.decode().splitlines() if "queries:" in l]
        print("N=%d Q=1M %s rc=%d %.2f s rows=%d | %s" % (n, " ".join(flags) or "(best hit)", r.returncode, dt, sum(1 for _ in open("/tmp/mo.tsv", "rb")), dbg), flush=True)
