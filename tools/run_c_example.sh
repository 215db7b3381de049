# Builds examples/decode_file.c and checks its output against the Python path (GPU box; run from a plain shell).
cd $GRAFT_REPO_ROOT
gcc -std=c99 -Wall -I include examples/decode_file.c -L zune-jpeg_amd -lzjhip -Wl,-rpath,$PWD/zune-jpeg_amd -o /tmp/decode_file || exit 1
/tmp/decode_file tests/golden/test-baseline.jpg /tmp/out.ppm || exit 1
/tmp/decode_file tests/golden/test-baseline.jpg /tmp/out_gpu.ppm gpu || exit 1
cmp /tmp/out.ppm /tmp/out_gpu.ppm && echo "the device entropy stage writes the same file"
python - <<'PY'
import importlib, numpy as np
zj = importlib.import_module("zune-jpeg_amd")
raw = open("/tmp/out.ppm", "rb").read()
header, body = raw.split(b"\n255\n", 1)
exp = zj.Decoder().decode_buffer(open("tests/golden/test-baseline.jpg", "rb").read())
print("header", header, "identical to the Python path:", np.array_equal(np.frombuffer(body, np.uint8), exp))
PY
