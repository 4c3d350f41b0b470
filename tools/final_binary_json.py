"""profiles/<round>_final_binary.json: the identity of the library the last whole `-m gpu` suite ran on (its sha256 and the hash of the sources
it is built from -- hipcc objects are not bit-reproducible, the source hash is what tests/test_measurement_tools.py holds the tree to).
    python tools/final_binary_json.py <pytest.log of the whole suite> <suite_log path under profiles/> <out.json>"""
import hashlib, json, os, re, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

log = open(sys.argv[1]).read()
m = re.findall(r"^(?:=+ )?(\d+ passed.*?)(?: =+)?$", log, re.M)
assert m and "failed" not in m[-1], "the suite log does not end green"
lib = os.path.join(bench.ROOT, "pointcloud_rl_amd", "libpcrl_hip.so")
rec = {"library_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(), "library_source_sha256": bench.library_source_sha(),
       "suite": m[-1], "suite_log": sys.argv[2],
       "note": "The last whole `python -m pytest tests -m gpu -x -q` ran on a library built from exactly these sources (make, sha256, suite in one "
               "visit). hipcc objects are not bit-reproducible: a rebuilt library has another sha256 and the same library_source_sha256."}
open(sys.argv[3], "w").write(json.dumps(rec, indent=1) + "\n")
print(json.dumps(rec))
