"""The HDF5 reader under AddressSanitizer + UBSan (SURVEY.md 5 "sanitizers"): a standalone C++
driver (waveformml_amd/csrc/h5_sanitize.cpp, `make -C waveformml_amd/csrc asan`) walks every fixture under
tests/golden/h5/ through every entry point of include/wfh5.h, then truncated copies and copies with flipped payload
bytes (raw gzip chunks, contiguous data blocks: what h5reader.cpp preads, inflates and converts itself).  Damaged
files must be REFUSED (WFH5_EIO / WFH5_EFORMAT / WFH5_EINVAL) or read as garbage values -- never crash, never trip a
sanitizer.  No Python in the sanitized process: ASan cannot be preloaded under this image's interpreter."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_h5_reader_survives_damaged_files_under_asan_and_ubsan(tmp_path):
    csrc = os.path.join(ROOT, "waveformml_amd", "csrc")
    subprocess.check_call(["make", "-C", csrc, "asan"], stdout=subprocess.DEVNULL)
    exe = os.path.join(ROOT, "waveformml_amd", "lib", "h5_sanitize_asan")
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    p = subprocess.run([exe, os.path.join(ROOT, "tests", "golden", "h5"), str(tmp_path)], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, (p.returncode, p.stdout[-2000:], p.stderr[-6000:])
    assert "h5_sanitize:" in p.stdout and "damaged copies" in p.stdout, p.stdout[-500:]
    # the walk really read things, and really met refusals
    line = [l for l in p.stdout.splitlines() if l.startswith("h5_sanitize:")][-1]
    ok = int(line.split("calls (")[1].split(" ok")[0])
    refused = int(line.split(" ok, ")[1].split(" refused")[0])
    assert ok > 1000 and refused > 1000, line
