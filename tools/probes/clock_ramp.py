"""Samples shader clock / package power / junction temperature (rocm-smi) while bench.py runs 12 000 K1 steps.
Result on MI355X: sclk reads 2.40 GHz from the first sample of the run to the last at ~870 W, 46 -> 50 C -- the step rate's
rise over a long run (1 018 steps/s over steps 6-25, 1 134 over steps 5 000-5 020) is NOT a clock ramp: the encoder backward
gets cheaper as training on the synthetic batch proceeds (fewer distinct argmax points per cloud: 228 -> 161 us), the
forward's duration does not move.   python tools/probes/clock_ramp.py"""
import os, re, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12000", "--warmup", "5",
                      "--no-cpu-baseline", "--no-experimental"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
t0 = time.time()
while p.poll() is None:
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--showtemp"], capture_output=True, text=True).stdout
    sclk = re.findall(r"sclk clock level.*?\((\d+)Mhz\)", out)
    pw = re.findall(r"Package Power \(W\): ([\d.]+)", out) or re.findall(r"Power \(W\): ([\d.]+)", out)
    tj = re.findall(r"junction\) \(C\): ([\d.]+)", out)
    print(f"t={time.time() - t0:5.1f}s sclk={sclk[:1]} power={pw[:1]} Tj={tj[:1]}", flush=True)
    time.sleep(0.7)
print(p.stdout.read()[-300:])
