"""A stand-in `ffmpeg` executable for the sink tests: write_stub(dir) puts a script named `ffmpeg` there that records its argv
(JSON) and every byte it reads from stdin next to the output path it was given, then exits with $DRBA_FFMPEG_STUB_RC (default 0).
The image has no ffmpeg; the tests put the stub first on PATH so that VideoFI_IO takes its encoder-pipe branch."""
import json
import os
import stat
import sys

_SCRIPT = """#!%s
import json, os, sys
out = sys.argv[-1]
with open(out + ".argv.json", "w") as f:
    json.dump(sys.argv, f)
rc = int(os.environ.get("DRBA_FFMPEG_STUB_RC", "0"))
limit = int(os.environ.get("DRBA_FFMPEG_STUB_READ", "-1"))
with open(out + ".stdin.bin", "wb") as f:
    got = 0
    while limit < 0 or got < limit:
        b = sys.stdin.buffer.read(1 << 20 if limit < 0 else min(1 << 20, limit - got))
        if not b:
            break
        f.write(b)
        got += len(b)
with open(out, "wb") as f:
    f.write(b"stub-container")
sys.exit(rc)
"""


def write_stub(directory):
    path = os.path.join(str(directory), "ffmpeg")
    with open(path, "w") as f:
        f.write(_SCRIPT % sys.executable)
    os.chmod(path, os.stat(path).st_mode | stat.S_IXUSR | stat.S_IXGRP | stat.S_IXOTH)
    return path


def recorded(output_path):
    """-> (argv list, stdin bytes) the stub saw for this output."""
    with open(output_path + ".argv.json") as f:
        argv = json.load(f)
    with open(output_path + ".stdin.bin", "rb") as f:
        data = f.read()
    return argv, data


def split_cmd(argv):
    """ffmpeg argv -> (input sections [[opts..., '-i', src], ...], output option dict, [maps], output path); options keep
    their value, the order of output options does not matter to ffmpeg."""
    assert os.path.basename(argv[0]) == "ffmpeg"
    body, out = argv[1:-1], argv[-1]
    inputs, cur, i = [], [], 0
    glob = []
    while i < len(body) and body[i] == "-y":
        glob.append(body[i])
        i += 1
    last_i = max(k for k, a in enumerate(body) if a == "-i")
    while i <= last_i + 1:
        cur.append(body[i])
        if body[i - 1] == "-i" and len(cur) >= 2:
            inputs.append(cur)
            cur = []
        i += 1
    rest = body[last_i + 2:]
    opts, maps = {}, []
    k = 0
    while k < len(rest):
        assert rest[k].startswith("-"), rest
        if rest[k] == "-map":
            maps.append(rest[k + 1])
        else:
            opts[rest[k]] = rest[k + 1]
        k += 2
    return glob, inputs, opts, maps, out
