"""`hc-edgecalc --resident` without a GPU: the launcher forwards the command line to a resident process (started on first use as a fresh
child of the launcher, which links neither libhcedge nor the HIP runtime) and relays its output and exit code; the parts of the program
that need no device — --help, the reference's argument checks (src/ViralQuasispecies.cpp:103-154), a missing FASTQ file — behave as in a
process of their own.  The stage itself through the resident process: tests/test_gpu_resident.py."""
import os
import subprocess
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "haploconduct_amd", "csrc", "hc-edgecalc")


VISIBILITY = ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "GPU_DEVICE_ORDINAL")


def _base_env(**kw):
    """The test's environment without device-visibility variables: they are part of a resident process's identity (a sub-directory per
    setting, test_device_visibility_is_part_of_the_resident_process_identity), the other tests look for the socket in the directory itself."""
    env = {k: v for k, v in os.environ.items() if k not in VISIBILITY}
    env.update(kw)
    return env


@pytest.fixture
def resident_env(tmp_path):
    env = _base_env(HC_RESIDENT_DIR=str(tmp_path / "res"), HC_RESIDENT_IDLE_S="30")
    yield env
    subprocess.run([EXE, "--resident_stop"], env=env, timeout=30)


def _run(args, env, **kw):
    return subprocess.run([EXE] + args, env=env, capture_output=True, text=True, timeout=60, **kw)


def test_launcher_links_no_hip():
    out = subprocess.run(["ldd", EXE], capture_output=True, text=True).stdout
    assert "amdhip" not in out and "libhcedge" not in out, "the launcher must start without loading the HIP runtime"


def test_resident_answers_like_a_process_of_its_own(resident_env, tmp_path):
    cases = [["--help"], ["--nonsense"], ["--overlaps", "x"], ["--singles", "s.fastq", "--overlaps", "o.txt"],
             ["--singles", "s.fastq", "--overlaps", "o.txt", "--original_readcount", "5", "--add_duplicates=true", "--resolve_orientations=true"],
             ["--singles", str(tmp_path / "missing.fastq"), "--overlaps", "o.txt", "--original_readcount", "5", "--output", str(tmp_path) + "/"]]
    for args in cases:
        own = _run(args, resident_env, cwd=tmp_path)
        res = _run(["--resident"] + args, resident_env, cwd=tmp_path)
        assert (res.returncode, res.stdout, res.stderr) == (own.returncode, own.stdout, own.stderr), args
    assert os.path.exists(os.path.join(resident_env["HC_RESIDENT_DIR"], "sock"))
    st = os.stat(resident_env["HC_RESIDENT_DIR"])
    assert (st.st_mode & 0o077) == 0, "the socket's directory is the user's alone"


def test_one_resident_process_serves_many_calls_and_a_closed_pipe_does_not_end_it(resident_env, tmp_path):
    assert _run(["--resident", "--help"], resident_env).returncode == 0
    log = os.path.join(resident_env["HC_RESIDENT_DIR"], "resident.log")
    pid = open(os.path.join(resident_env["HC_RESIDENT_DIR"], "pid")).read()
    p = subprocess.Popen([EXE, "--resident", "--help"], env=resident_env, stdout=subprocess.PIPE)
    p.stdout.close()  # `| head -0`: the resident process writes into a closed pipe
    p.wait(timeout=30)
    for _ in range(5):
        assert _run(["--resident", "--help"], resident_env).returncode == 0
    assert open(os.path.join(resident_env["HC_RESIDENT_DIR"], "pid")).read() == pid, "the same resident process all along"
    assert os.path.exists(log)
    # the working directory of the CLIENT counts (relative --output)
    (tmp_path / "wd").mkdir()
    r = _run(["--resident", "--singles", "nope.fastq", "--overlaps", "o.txt", "--original_readcount", "3"], resident_env, cwd=tmp_path / "wd")
    assert r.returncode == 1 and "nope.fastq" in r.stderr
    assert (tmp_path / "wd" / "viralquasispecies.log").exists(), "the log goes where the client stands (default --output is the working directory)"


def test_stop_and_idle_time_out(tmp_path):
    env = _base_env(HC_RESIDENT_DIR=str(tmp_path / "res2"), HC_RESIDENT_IDLE_S="1")
    assert _run(["--resident", "--help"], env).returncode == 0
    sock = os.path.join(env["HC_RESIDENT_DIR"], "sock")
    assert os.path.exists(sock)
    for _ in range(100):  # leaves by itself after a second without a request
        if not os.path.exists(sock):
            break
        time.sleep(0.1)
    assert not os.path.exists(sock)
    assert _run(["--resident", "--help"], env).returncode == 0  # ... and comes back on the next call
    assert subprocess.run([EXE, "--resident_stop"], env=env, timeout=30).returncode == 0
    for _ in range(50):
        if not os.path.exists(sock):
            break
        time.sleep(0.1)
    assert not os.path.exists(sock)
    assert subprocess.run([EXE, "--resident_stop"], env=env, timeout=30).returncode == 0, "nobody to stop is not an error"


def test_a_rebuilt_library_retires_the_resident_process(resident_env, tmp_path):
    """A resident process that loaded ANOTHER libhcedge.so than the one next to the executable now (the library was rebuilt under it) must not
    answer for it: it says so and leaves, the client starts a new one.  Here: a copy of the executable + library in a directory of its own,
    the library's mtime bumped between two calls."""
    import shutil

    d = tmp_path / "bin"
    d.mkdir()
    exe, lib = str(d / "hc-edgecalc"), str(d / "libhcedge.so")
    shutil.copy2(EXE, exe)
    shutil.copy2(os.path.join(os.path.dirname(EXE), "libhcedge.so"), lib)
    try:
        assert subprocess.run([exe, "--resident", "--help"], env=resident_env, capture_output=True, timeout=60).returncode == 0
        pid = open(os.path.join(resident_env["HC_RESIDENT_DIR"], "pid")).read()
        os.utime(lib, (time.time() + 5, time.time() + 5))
        r = subprocess.run([exe, "--resident", "--help"], env=resident_env, capture_output=True, text=True, timeout=60)
        assert r.returncode == 0 and "Program options" in r.stdout
        assert open(os.path.join(resident_env["HC_RESIDENT_DIR"], "pid")).read() != pid, "a new resident process answers"
    finally:
        subprocess.run([exe, "--resident_stop"], env=resident_env, timeout=30)



def test_many_clients_at_once_start_one_resident_process(tmp_path):
    """Eight clients started together on a socket nobody serves yet: every one is answered, one resident process remains (the others that
    were started by the race leave as soon as they see the socket served)."""
    env = _base_env(HC_RESIDENT_DIR=str(tmp_path / "res4"), HC_RESIDENT_IDLE_S="30")
    try:
        ps = [subprocess.Popen([EXE, "--resident", "--help"], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE) for _ in range(8)]
        for p in ps:
            out, err = p.communicate(timeout=60)
            assert p.returncode == 0 and b"Program options" in out, err
        time.sleep(0.5)
        want = f"--resident_daemon\0{env['HC_RESIDENT_DIR']}/sock".encode()
        alive = 0
        for pid in os.listdir("/proc"):
            if pid.isdigit():
                try:
                    alive += want in open(f"/proc/{pid}/cmdline", "rb").read()
                except OSError:
                    pass
        assert alive == 1, f"{alive} resident processes on one socket"
    finally:
        subprocess.run([EXE, "--resident_stop"], env=env, timeout=30)


def test_a_client_whose_directory_is_gone_gets_an_exit_code_and_the_resident_process_lives(resident_env, tmp_path):
    """Round-5 advisor: chdir() into the client's working directory failing took the resident process down (a null reply handle) with the
    client's descriptors attached.  The client now gets exit code 1 and the message; the same resident process answers the next call."""
    assert _run(["--resident", "--help"], resident_env).returncode == 0
    pid = open(os.path.join(resident_env["HC_RESIDENT_DIR"], "pid")).read()
    gone = tmp_path / "gone"
    gone.mkdir()
    # the client stands in a directory that is removed before it calls (getcwd fails -> "/" is sent? no: the launcher reads its cwd first)
    code = ("import os, subprocess, sys; os.chdir(sys.argv[1]); os.rmdir(sys.argv[1]); "
            "r = subprocess.run([sys.argv[2], '--resident', '--help'], capture_output=True, text=True); "
            "print(r.returncode); sys.stderr.write(r.stderr)")
    r = subprocess.run(["python3", "-c", code, str(gone), EXE], env=resident_env, capture_output=True, text=True, timeout=60)
    # a removed directory: getcwd() fails in the launcher, which then sends "/" — the call succeeds from there; either way an answer comes back
    assert r.stdout.strip() in ("0", "1"), (r.stdout, r.stderr)
    # a directory the resident process cannot enter: a path that does not exist, sent by a client standing in a directory renamed under it
    moved = tmp_path / "here"
    moved.mkdir()
    code2 = ("import os, subprocess, sys; os.chdir(sys.argv[1]); "
             "real_getcwd = os.getcwd(); os.rename(sys.argv[1], sys.argv[1] + '.moved'); os.mkdir(sys.argv[1]); os.rmdir(sys.argv[1]); "
             "r = subprocess.run([sys.argv[2], '--resident', '--help'], capture_output=True, text=True); print(r.returncode); sys.stderr.write(r.stderr)")
    r2 = subprocess.run(["python3", "-c", code2, str(moved), EXE], env=resident_env, capture_output=True, text=True, timeout=60)
    assert r2.stdout.strip() in ("0", "1"), (r2.stdout, r2.stderr)
    assert "went away" not in r.stderr and "went away" not in r2.stderr
    assert _run(["--resident", "--help"], resident_env).returncode == 0
    assert open(os.path.join(resident_env["HC_RESIDENT_DIR"], "pid")).read() == pid, "the resident process survived both"


def test_unenterable_directory_is_answered_with_code_1(resident_env, tmp_path):
    """The resident loop's own branch: the request names a working directory that does not exist (sent here by hand, as the launcher would)."""
    import socket
    import struct

    assert _run(["--resident", "--help"], resident_env).returncode == 0
    pid = open(os.path.join(resident_env["HC_RESIDENT_DIR"], "pid")).read()
    lib = os.path.join(os.path.dirname(EXE), "libhcedge.so")
    st = os.stat(lib)
    stamp = (int(st.st_mtime_ns // 10**9) * 1000000007 + int(st.st_mtime_ns % 10**9) + (st.st_size << 20)) & (2**64 - 1)
    out_r, out_w = os.pipe()
    err_r, err_w = os.pipe()
    s = socket.socket(socket.AF_UNIX, socket.SOCK_STREAM)
    s.connect(os.path.join(resident_env["HC_RESIDENT_DIR"], "sock"))
    cwd = str(tmp_path / "no" / "such" / "dir").encode()
    args = [b"hc-edgecalc", b"--help"]
    head = struct.pack("<7I", 0x48435235, len(args), 0, len(cwd), 0, stamp & 0xFFFFFFFF, stamp >> 32)
    socket.send_fds(s, [head], [out_w, err_w])
    os.close(out_w)
    os.close(err_w)
    for b in [cwd] + args:
        s.sendall(struct.pack("<I", len(b)) + b)
    s.settimeout(30)
    code = s.recv(4)
    assert len(code) == 4 and struct.unpack("<i", code)[0] == 1
    assert b"cannot enter" in os.read(err_r, 4096)
    s.close()
    assert _run(["--resident", "--help"], resident_env).returncode == 0
    assert open(os.path.join(resident_env["HC_RESIDENT_DIR"], "pid")).read() == pid, "the resident process is still the same one"


def test_device_visibility_is_part_of_the_resident_process_identity(tmp_path):
    """Round-5 advisor: two clients with different HIP_VISIBLE_DEVICES must not share one resident process (which devices a process sees is
    fixed when its runtime starts).  Each setting gets a sub-directory — socket, lock, pid — of its own."""
    base = _base_env(HC_RESIDENT_DIR=str(tmp_path / "res5"), HC_RESIDENT_IDLE_S="30")
    e0, e1 = dict(base, HIP_VISIBLE_DEVICES="0"), dict(base, HIP_VISIBLE_DEVICES="1")
    try:
        for env in (base, e0, e1, e0):
            assert _run(["--resident", "--help"], env).returncode == 0
        subs = sorted(d for d in os.listdir(base["HC_RESIDENT_DIR"]) if d.startswith("vis-"))
        assert len(subs) == 2, subs
        pids = {open(os.path.join(base["HC_RESIDENT_DIR"], d, "pid")).read() for d in subs}
        pids.add(open(os.path.join(base["HC_RESIDENT_DIR"], "pid")).read())
        assert len(pids) == 3, "three settings, three resident processes"
    finally:
        for env in (base, e0, e1):
            subprocess.run([EXE, "--resident_stop"], env=env, timeout=30)


def test_socket_directory_must_not_be_a_symbolic_link(tmp_path):
    real = tmp_path / "real"
    real.mkdir(mode=0o700)
    link = tmp_path / "link"
    os.symlink(real, link)
    env = _base_env(HC_RESIDENT_DIR=str(link))
    r = _run(["--resident", "--help"], env)
    assert r.returncode == 1 and "not a link" in r.stderr
    assert os.listdir(real) == []
