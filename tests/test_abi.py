"""CPU: the C-ABI library loads, exports exactly the entry points include/customnerf_hip.h declares, the ctypes binding
lists the same set, and argument validation that happens before any launch behaves as documented.
No GPU compute is issued here."""
import ctypes
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADERS = [os.path.join(ROOT, "include", "customnerf_hip.h"), os.path.join(ROOT, "include", "customnerf_sd.h")]


def header_source():
    return re.sub(r"/\*.*?\*/", "", "\n".join(open(h).read() for h in HEADERS), flags=re.S)


def declared_symbols():
    return sorted(set(re.findall(r"\b(cnerf_[A-Za-z0-9_]+)\s*\(", header_source())))


def test_header_symbols_are_exported_and_bound():
    from customnerf_amd import _lib
    decl = declared_symbols()
    assert len(decl) >= 20
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in decl:
        assert hasattr(lib, name), f"{name} declared in the header but not exported"
    bound = set(_lib.SIGNATURES) | {"cnerf_target_arch"}
    assert bound == set(decl), f"binding/header mismatch: {bound ^ set(decl)}"
    exported = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH], text=True)
    exported = sorted(set(re.findall(r"\bT (cnerf_[A-Za-z0-9_]+)", exported)))
    assert exported == decl, f"library exports differ from the header: {set(exported) ^ set(decl)}"


def test_library_identity_and_arity():
    from customnerf_amd import _lib
    assert _lib.lib.cnerf_abi_version() == _lib.ABI_VERSION
    assert _lib.lib.cnerf_target_arch() == b"gfx950"
    src = header_source()
    for name, argtypes in _lib.SIGNATURES.items():
        m = re.search(r"\b%s\s*\(([^;]*?)\)\s*;" % name, src, flags=re.S)
        assert m, name
        args = m.group(1).strip()
        n = 0 if args in ("", "void") else len(args.split(","))
        assert n == len(argtypes), f"{name}: header has {n} parameters, binding {len(argtypes)}"


def test_code_object_is_gfx950_only():
    """The fat binary embedded in the .so targets gfx950 and nothing else (no multi-arch / CUDA dual paths)."""
    from customnerf_amd import _lib
    blob = open(_lib.LIB_PATH, "rb").read()
    targets = set(re.findall(rb"amdgcn-amd-amdhsa--(gfx[0-9a-f]+)", blob))
    assert targets == {b"gfx950"}, targets


def test_no_binaries_tracked_in_git():
    """History stays source-only: nothing under customnerf_amd/ that git tracks is an ELF / code-object / offload bundle."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    if not os.path.isdir(os.path.join(root, ".git")):
        pytest.skip("not a git checkout (GPU-box snapshot)")
    files = subprocess.run(["git", "ls-files", "customnerf_amd", "oracle", "include"], cwd=root, capture_output=True, text=True, check=True).stdout.split()
    assert files
    bad = []
    for f in files:
        p = os.path.join(root, f)
        if not os.path.isfile(p):
            continue                      # deleted in the work tree, not yet committed
        head = open(p, "rb").read(24)
        if head.startswith(b"\x7fELF") or head.startswith(b"__CLANG_OFFLOAD_BUNDLE__") or os.path.getsize(p) == 0 and ".so." in f:
            bad.append(f)
    assert not bad, bad


def test_argument_validation_without_launch():
    """Rejected arguments return before anything touches the device (safe on a CPU-only box)."""
    from customnerf_amd._lib import lib
    off = np.array([0, 8, 16], np.int32)
    one = ctypes.c_void_p(16)           # non-NULL dummy, never dereferenced on these paths
    # C = 3 unsupported (gridencoder.cu:380), D = 6 unsupported (:397), L > 32, bad gridtype
    assert lib.cnerf_grid_encode_forward(one, one, off.ctypes.data, one, 10, 3, 3, 2, 2, 1.0, 16, None, 0, 0, 0, 0, None) == -1
    assert lib.cnerf_grid_encode_forward(one, one, off.ctypes.data, one, 10, 6, 2, 2, 2, 1.0, 16, None, 0, 0, 0, 0, None) == -1
    assert lib.cnerf_grid_encode_forward(one, one, off.ctypes.data, one, 10, 3, 2, 40, 40, 1.0, 16, None, 0, 0, 0, 0, None) == -1
    assert lib.cnerf_grid_encode_forward(one, one, off.ctypes.data, one, 10, 3, 2, 2, 2, 1.0, 16, None, 7, 0, 0, 0, None) == -1
    assert lib.cnerf_grid_encode_forward(one, one, off.ctypes.data, one, 10, 3, 2, 2, 2, 1.0, 16, None, 0, 0, 0, 9, None) == -1
    assert lib.cnerf_grid_encode_forward(None, one, off.ctypes.data, one, 10, 3, 2, 2, 2, 1.0, 16, None, 0, 0, 0, 0, None) == -2
    assert lib.cnerf_grid_encode_forward(one, one, None, one, 10, 3, 2, 2, 2, 1.0, 16, None, 0, 0, 0, 0, None) == -2
    bad = np.array([0, 8, 8], np.int32)  # empty level
    assert lib.cnerf_grid_encode_forward(one, one, bad.ctypes.data, one, 10, 3, 2, 2, 2, 1.0, 16, None, 0, 0, 0, 0, None) == -1
    assert lib.cnerf_grid_encode_backward(one, one, off.ctypes.data, None, 10, 3, 2, 2, 2, 1.0, 16, None, None, 0, 0, 0, 0, None, 0, None) == -2
    # workspace query: small problems / unsupported shapes -> 0 bytes (atomic path); the 128x128x128-sample step -> ~2 GiB of records
    need = ctypes.c_uint64(123)
    big = np.array([0, 4920, 18744, 51512, 136696, 352696, 876984] + [876984 + 524288 * i for i in range(1, 11)], np.int32)
    S = float(np.log2(np.exp2(np.log2(2048 / 16) / 15)))
    assert lib.cnerf_grid_encode_backward_workspace_bytes(big.ctypes.data, 1000, 3, 2, 16, 16, S, 16, 1, ctypes.addressof(need)) == 0 and need.value == 0
    assert lib.cnerf_grid_encode_backward_workspace_bytes(big.ctypes.data, 2097152, 3, 4, 16, 16, S, 16, 1, ctypes.addressof(need)) == 0 and need.value == 0
    assert lib.cnerf_grid_encode_backward_workspace_bytes(big.ctypes.data, 2097152, 3, 2, 16, 16, S, 16, 1, ctypes.addressof(need)) == 0
    # round 5 (no histogram pre-pass): the point blocks' private regions (worst case: every record) + the hashed bins' regions (1.5 x the expected
    # load) + run table / prefix + fixed-point partial images of split bins
    assert 2097152 * 16 * 8 * 8 <= need.value < 2097152 * 16 * 8 * 8 * 1.85
    assert lib.cnerf_grid_encode_backward_workspace_bytes(big.ctypes.data, 2097152, 3, 2, 16, 16, S, 16, 0, ctypes.addressof(need)) == 0
    assert 2097152 * 16 * 8 * 12 <= need.value < 2097152 * 16 * 8 * 12 * 1.05
    # empty work is accepted without a launch
    assert lib.cnerf_grid_encode_forward(one, one, off.ctypes.data, one, 0, 3, 2, 2, 2, 1.0, 16, None, 0, 0, 0, 0, None) == 0
    assert lib.cnerf_near_far_from_aabb(one, one, one, 0, 0.1, one, one, None) == 0
    assert lib.cnerf_near_far_from_aabb(None, one, one, 4, 0.1, one, one, None) == -2
    assert lib.cnerf_composite_rays_train_forward(one, one, one, one, 4, 4, 1e-4, one, one, one, 5, None) == -1   # rgb stride 5
    assert lib.cnerf_march_rays_train_count(one, one, one, 2.0, 0.0, 1024, 4, 0, 128, one, one, one, one, one, None) == -1   # C = 0
    assert lib.cnerf_adam_step(one, one, one, one, None, 8, 1e-3, 0.9, 0.99, 1e-15, 0, 1.0, 1, None) == -1                   # step 0
    assert lib.cnerf_generate_rays(one, 1, 4, 4, 0.0, 1.0, 2.0, 2.0, 1.0, 0, one, one, None) == -1                             # fx = 0
    # round 5 entry points: the edit step's single-launch glue and the plan query
    assert lib.cnerf_edit_ray_images(None, 1, 16, one, one, one, None) == -2 and lib.cnerf_edit_ray_images(one, 0, 16, one, one, one, None) == -1
    assert lib.cnerf_edit_ray_images_backward(None, None, None, 1, 16, None, None) == -2           # the three gradients may be NULL, the output may not
    assert lib.cnerf_edit_l1_loss(one, one, 0, 1.0, one, one, None) == -1 and lib.cnerf_edit_l1_loss(one, None, 8, 1.0, one, one, None) == -2
    assert lib.cnerf_edit_sds_loss(one, one, 0, one, one, None) == -1 and lib.cnerf_edit_sds_loss(None, one, 8, one, one, None) == -2
    assert lib.cnerf_edit_scale_by_scalar(one, one, 1.0, 0, one, None) == 0 and lib.cnerf_edit_scale_by_scalar(one, None, 1.0, 8, one, None) == -2
    assert lib.cnerf_sd_sample_latents(one, one, 0, 64, 0.18215, one, None) == -1 and lib.cnerf_sd_sample_latents(None, one, 1, 64, 0.18215, one, None) == -2
    assert lib.cnerf_sd_sample_latents_backward(one, one, None, 1, 64, 0.18215, one, None) == -2
    vals = (ctypes.c_float * 2)(1.0, 2.0)
    assert lib.cnerf_set_floats(one, ctypes.cast(vals, ctypes.c_void_p), 17, None) == -1 and lib.cnerf_set_floats(None, ctypes.cast(vals, ctypes.c_void_p), 2, None) == -2
    needs = ctypes.c_int(7)
    assert lib.cnerf_grid_encode_backward_needs_plan(big.ctypes.data, 2097152, 3, 2, 16, 16, S, 16, 0, 1, ctypes.addressof(needs)) == 0 and needs.value == 0   # third form: no plan
    assert lib.cnerf_grid_encode_backward_needs_plan(big.ctypes.data, 2097152, 3, 2, 16, 16, S, 16, 0, 0, ctypes.addressof(needs)) == 0 and needs.value == 1   # float32 records: second form


def test_host_side_modules_on_cpu():
    """Module construction (offset tables, parameter shapes/names, tcnn parameter counts) needs no GPU;
    evaluating on CPU tensors must fail loudly (no CPU path)."""
    import torch
    from customnerf_amd.gridencoder import GridEncoder
    from customnerf_amd.nerf.network_grid import NeRFNetwork
    from customnerf_amd.scene import make_opt
    from customnerf_amd import tcnn
    g = np.load(os.path.join(ROOT, "tests", "golden", "grid_offsets.npz"))
    for tag, kw in (("hash_L16_T19_2048", dict(num_levels=16, log2_hashmap_size=19, desired_resolution=2048, gridtype='hash')),
                    ("tiled_L16_T21_8192", dict(num_levels=16, log2_hashmap_size=21, desired_resolution=8192, gridtype='tiled')),
                    ("hash_L4_T19_2048", dict(num_levels=4, log2_hashmap_size=19, desired_resolution=2048, gridtype='hash')),
                    ("hash_default", dict())):
        enc = GridEncoder(**kw)
        np.testing.assert_array_equal(enc.offsets.numpy(), g[tag + "__offsets"])          # reference's own table
        assert float(enc.per_level_scale) == float(g[tag + "__pls"])
        assert tuple(enc.embeddings.shape) == tuple(g[tag + "__shape"])
        assert int(enc.n_params) == int(g[tag + "__n_params"]) and enc.output_dim == int(g[tag + "__output_dim"])
        assert float(enc.embeddings.detach().abs().max()) <= 1e-4
    with pytest.raises(RuntimeError):
        enc(torch.rand(4, 3))
    assert tcnn.Network(32, 64, {"n_neurons": 64, "n_hidden_layers": 2}).params.numel() == 10240
    assert tcnn.Network(64, 1, {"n_neurons": 64, "n_hidden_layers": 1}).params.numel() == 5120
    assert tcnn.Network(91, 4, {"n_neurons": 64, "n_hidden_layers": 1, "output_activation": "Sigmoid"}).params.numel() == 7168
    with pytest.raises(ValueError):
        tcnn.Network(32, 64, {"n_neurons": 128})
    model = NeRFNetwork(make_opt(cuda_ray=True))
    keys = set(model.state_dict().keys())
    for k in ("aabb_train", "aabb_infer", "pos_en.embeddings", "pos_en.offsets", "network.params", "density_network.params",
              "rgb_network.params", "density_grid", "density_bitfield", "step_counter"):
        assert k in keys, k                                                                # SURVEY.md §5 checkpoint key set
    assert model.cascade == 2 and model.density_bitfield.numel() == 2 * 128 ** 3 // 8
    assert sum(p.numel() for n, p in model.named_parameters() if 'pos_en' not in n) == 22528
    groups = model.get_params(5e-4)
    assert groups[0]['lr'] == 5e-3 and len(groups) == 4
