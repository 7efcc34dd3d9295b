"""The Rust binding lives in INTEGRATION.md (the in-tree arm a maintainer would add) and in shim/rustfft-mi355 (a standalone
crate over the public `rustfft::Fft` trait) as source (no rustc / cargo in this image or on the GPU
box, so it cannot be compiled here): this test keeps it honest against include/mi355fft.h -- every `extern "C"` function the
document declares must exist in the header with the same arity, the same pointer-ness / constness and the same scalar
types; every mi355fft_* function the Rust snippets CALL must be declared in one of the extern blocks; and the #[repr(C)]
Mi355PlanOptions must list the header's mi355fft_plan_options fields in the same order with matching types."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

C_TO_RUST = {
    "int": "c_int", "size_t": "usize", "double": "f64", "void*": "*mut c_void", "const void*": "*const c_void",
    "mi355fft_plan**": "*mut *mut Mi355Plan", "mi355fft_plan*": "*mut Mi355Plan", "const mi355fft_plan*": "*const Mi355Plan",
    "const mi355fft_plan_options*": "*const Mi355PlanOptions", "const char*": "*const std::ffi::c_char", "char*": "*mut std::ffi::c_char",
    "double*": "*mut f64", "size_t*": "*mut usize",
    "mi355fft_twiddle_fn": "Option<extern \"C\" fn(*mut c_void, usize, usize, *mut f64, *mut f64)>",
    "const mi355fft_recipe_node*": "*const Mi355RecipeNode", "float*": "*mut f32",
    # round 4: the multi-device plan and the fused-launch hooks
    "mi355fft_multi_plan**": "*mut *mut Mi355MultiPlan", "mi355fft_multi_plan*": "*mut Mi355MultiPlan", "const mi355fft_multi_plan*": "*const Mi355MultiPlan",
    "const int*": "*const c_int", "void*const*": "*const *mut c_void", "const void*const*": "*const *const c_void", "unsigned*": "*mut c_uint",
}


def norm_c(t):
    t = re.sub(r"\s+", " ", t.strip())
    t = re.sub(r"\s*\*\s*", "*", t)
    return t


def c_prototypes(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"^([A-Za-z_][\w\s\*]*?)\b(mi355fft_\w+)\s*\(([^;{}]*)\)\s*;", text, flags=re.M):
        ret, name, args = norm_c(m.group(1)), m.group(2), m.group(3).strip()
        params = []
        if args and args != "void":
            for a in args.split(","):
                a = norm_c(a)
                mm = re.match(r"^(.*?[\*\s])(\w+)$", a)  # strip the parameter name
                params.append(norm_c(mm.group(1)) if mm else a)
        protos[name] = (ret, params)
    return protos


def rust_externs(text):
    fns = {}
    for block in re.findall(r'extern "C" \{(.*?)\n\}', text, flags=re.S):
        for m in re.finditer(r"fn (mi355fft_\w+)\s*\((.*?)\)\s*(?:->\s*([^;]+))?;", block, flags=re.S):
            args = [re.sub(r"\s+", " ", a.split(":", 1)[1].strip()) for a in re.split(r",(?![^<(]*[>)])", m.group(2)) if a.strip()]
            fns[m.group(1)] = ((m.group(3) or "()").strip(), args)
    return fns


def test_extern_block_matches_the_header():
    header = open(os.path.join(ROOT, "include", "mi355fft.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    protos, externs = c_prototypes(header), rust_externs(doc)
    assert len(externs) >= 11, sorted(externs)
    for name, (rret, rargs) in externs.items():
        assert name in protos, f"INTEGRATION.md declares {name}, which include/mi355fft.h does not"
        cret, cargs = protos[name]
        assert C_TO_RUST[cret] == rret, (name, cret, rret)
        assert len(cargs) == len(rargs), (name, cargs, rargs)
        for ca, ra in zip(cargs, rargs):
            assert C_TO_RUST[ca] == ra, (name, ca, ra)
    # every entry point the Rust snippets call is declared
    rust = "\n".join(re.findall(r"```rust(.*?)```", doc, flags=re.S))
    for used in set(re.findall(r"\b(mi355fft_\w+)\s*\(", rust)):
        assert used in externs, f"the Rust snippets call {used} without declaring it in an extern block"


def test_plan_options_struct_matches_the_header():
    header = open(os.path.join(ROOT, "include", "mi355fft.h")).read()
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    cbody = re.search(r"typedef struct mi355fft_plan_options \{(.*?)\} mi355fft_plan_options;", header, flags=re.S).group(1)
    cfields = []
    for line in cbody.split(";"):
        line = norm_c(line)
        if line:
            mm = re.match(r"^(.*?[\*\s])(\w+)$", line)
            cfields.append((mm.group(2), norm_c(mm.group(1))))
    rbody = re.search(r"#\[repr\(C\)\]\s*struct Mi355PlanOptions \{(.*?)\n\}", doc, flags=re.S).group(1)
    rbody = re.sub(r"//.*", "", rbody)
    rfields = [(a.split(":", 1)[0].strip(), re.sub(r"\s+", " ", a.split(":", 1)[1].strip())) for a in re.split(r",(?![^<(]*[>)])", rbody) if a.strip()]
    assert [f for f, _ in cfields] == [f for f, _ in rfields], (cfields, rfields)
    for (cn, ct), (rn, rt) in zip(cfields, rfields):
        assert C_TO_RUST[ct] == rt, (cn, ct, rt)


# ---- shim/rustfft-mi355: the standalone crate ------------------------------------------------------------------------------
CRATE = os.path.join(ROOT, "shim", "rustfft-mi355")


def _strip_rust(text):
    text = re.sub(r"//[^\n]*", "", text)
    return re.sub(r"\bpub\s+", "", text)


def _rust_struct_fields(text, name):
    body = re.search(r"#\[repr\(C\)\](?:\s*#\[[^\]]*\])*\s*struct %s \{(.*?)\n\s*\}" % name, text, flags=re.S).group(1)
    return [(a.split(":", 1)[0].strip(), re.sub(r"\s+", " ", a.split(":", 1)[1].strip())) for a in re.split(r",(?![^<(]*[>)])", body) if a.strip()]


def _c_struct_fields(header, name):
    body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (name, name), re.sub(r"/\*.*?\*/", "", header, flags=re.S), flags=re.S).group(1)
    fields = []
    for line in body.split(";"):
        line = norm_c(line)
        if line:
            mm = re.match(r"^(.*?[\*\s])(\w+)$", line)
            fields.append((mm.group(2), norm_c(mm.group(1))))
    return fields


def test_crate_binds_every_function_of_the_header():
    header = open(os.path.join(ROOT, "include", "mi355fft.h")).read()
    src = _strip_rust(open(os.path.join(CRATE, "src", "lib.rs")).read())
    protos, externs = c_prototypes(header), rust_externs(src)
    assert set(externs) == set(protos), (sorted(set(protos) - set(externs)), sorted(set(externs) - set(protos)))
    for name, (rret, rargs) in externs.items():
        cret, cargs = protos[name]
        assert C_TO_RUST[cret] == rret, (name, cret, rret)
        assert len(cargs) == len(rargs), (name, cargs, rargs)
        for ca, ra in zip(cargs, rargs):
            assert C_TO_RUST[ca] == ra, (name, ca, ra)
    # every ffi:: call in the crate (and its tests) names a declared function
    used = set()
    for root, _dirs, files in os.walk(CRATE):
        for f in files:
            if f.endswith(".rs"):
                used |= set(re.findall(r"ffi::(mi355fft_\w+)", open(os.path.join(root, f)).read()))
    assert used and used <= set(externs), sorted(used - set(externs))


def test_crate_structs_and_constants_match_the_header():
    header = open(os.path.join(ROOT, "include", "mi355fft.h")).read()
    src = _strip_rust(open(os.path.join(CRATE, "src", "lib.rs")).read())
    for cname, rname in (("mi355fft_plan_options", "Mi355PlanOptions"), ("mi355fft_recipe_node", "Mi355RecipeNode")):
        cfields, rfields = _c_struct_fields(header, cname), _rust_struct_fields(src, rname)
        assert [f for f, _ in cfields] == [f for f, _ in rfields], (cfields, rfields)
        for (cn, ct), (rn, rt) in zip(cfields, rfields):
            assert C_TO_RUST[ct] == rt, (cn, ct, rt)
    cvals = {k: int(v) for k, v in re.findall(r"\b(MI355FFT_\w+)\s*=\s*(-?\d+)", header)}
    for rname, val in re.findall(r"const (\w+): c_int = (-?\d+);", src):
        assert cvals["MI355FFT_" + rname] == int(val), rname
    recipe_kinds = {k for k in cvals if k.startswith("MI355FFT_RECIPE_") and not k.startswith("MI355FFT_RECIPE_STATUS_")}
    assert {"MI355FFT_" + r for r, _ in re.findall(r"const (RECIPE_\w+): c_int = (\d+);", src)} == recipe_kinds


def test_crate_layout():
    for rel in ("Cargo.toml", "build.rs", "src/lib.rs", "tests/accuracy.rs", "tests/host_planner.rs"):
        assert os.path.isfile(os.path.join(CRATE, rel)), rel
    manifest = open(os.path.join(CRATE, "Cargo.toml")).read()
    assert re.search(r'^rustfft\s*=', manifest, flags=re.M) and "[features]" in manifest


def _split_args(text):
    args, depth, cur = [], 0, ""
    for ch in text:
        if ch in "(<[":
            depth += 1
        elif ch in ")>]":
            depth -= 1
        if ch == "," and depth == 0:
            args.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        args.append(cur.strip())
    return args


def test_crate_call_sites_pass_the_arguments_in_the_headers_order():
    """Declarations that match the header are not enough: a call site can still swap two arguments of the same type (input and
    output pointers, or a length with the other buffer's).  Every trait method of `impl Fft<T>` (one-device and multi-device
    object) must call ITS entry point -- process_with_scratch -> *_inplace_host, process_outofplace_with_scratch ->
    *_outofplace_host, process_immutable_with_scratch -> *_immutable_host -- with (plan, pointer of X, X.len(), ...) in the order
    of the header's parameter list, const-ness of the pointer included."""
    header = open(os.path.join(ROOT, "include", "mi355fft.h")).read()
    protos = c_prototypes(header)
    src = re.sub(r"//[^\n]*", "", open(os.path.join(CRATE, "src", "lib.rs")).read())
    want = {
        "process_with_scratch": ("inplace", [("plan",), ("mut", "buffer"), ("len", "buffer"), ("mut", "scratch"), ("len", "scratch")]),
        "process_outofplace_with_scratch": ("outofplace", [("plan",), ("mut", "input"), ("len", "input"), ("mut", "output"), ("len", "output"), ("mut", "scratch"), ("len", "scratch")]),
        "process_immutable_with_scratch": ("immutable", [("plan",), ("const", "input"), ("len", "input"), ("mut", "output"), ("len", "output"), ("mut", "scratch"), ("len", "scratch")]),
    }
    checked = 0
    for impl_for, prefix in (("HipFft", "mi355fft_process_"), ("HipFftMulti", "mi355fft_multi_process_")):
        body = re.search(r"impl<T: FftNum> Fft<T> for %s<T> \{(.*?)\n    \}\n" % impl_for, src, flags=re.S).group(1)
        for method, (mode, roles) in want.items():
            fn_body = re.search(r"fn %s\(.*?\) \{(.*?)\n        \}" % method, body, flags=re.S).group(1)
            calls = re.findall(r"ffi::(mi355fft_\w+)\s*\((.*?)\)\s*\n?\s*\};", fn_body, flags=re.S)
            assert len(calls) == 1, (impl_for, method, calls)
            name, argtext = calls[0]
            assert name == prefix + mode + "_host", (impl_for, method, name)
            args = _split_args(argtext)
            cparams = protos[name][1]
            assert len(args) == len(cparams) == len(roles), (name, args, cparams)
            for arg, role, ctype in zip(args, roles, cparams):
                if role[0] == "plan":
                    assert arg == "self.plan", (name, arg)
                elif role[0] == "len":
                    assert arg == role[1] + ".len()" and ctype == "size_t", (name, arg, ctype)
                elif role[0] == "mut":
                    assert arg == role[1] + ".as_mut_ptr() as *mut c_void" and ctype == "void*", (name, arg, ctype)
                else:
                    assert arg == role[1] + ".as_ptr() as *const c_void" and ctype == "const void*", (name, arg, ctype)
            checked += 1
    assert checked == 6
    # the header's own parameter NAMES carry the same order (buffer / n_elems; input, n_in, output, n_out)
    flat = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    for name in ("mi355fft_process_outofplace_host", "mi355fft_multi_process_outofplace_host", "mi355fft_process_immutable_host", "mi355fft_multi_process_immutable_host"):
        params = re.search(name + r"\s*\((.*?)\)\s*;", flat, flags=re.S).group(1)
        names = [re.sub(r"\s+", " ", a).strip().split(" ")[-1].lstrip("*") for a in params.split(",")]
        assert names == ["plan", "input", "n_in", "output", "n_out", "scratch", "scratch_elems"], (name, names)
