"""Kernel names as rocprofv3 reports them -> the kernel's full name with its template arguments, without its parameter list
(shared by tools/pmc_summary.py and tools/pmc_traffic.py)."""
_demangled = {}


def kernel_name(raw):
    """The kernel's FULL name with its template arguments and without its parameter list: `repet::(anonymous
    namespace)::stft_reg_kernel<2>` and `...::istft_ola_reg_kernel<2, 0>` are different kernels and stay apart (round 4's
    summary cut every name at the first '(' -- the one of "(anonymous namespace)" -- and lumped six kernels into "repet::")."""
    if raw in _demangled:
        return _demangled[raw]
    name = raw
    if name.startswith("_Z"):                               # rocprofv3 leaves some names mangled
        import re
        import subprocess
        for tool in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):      # (GNU c++filt does not know _Float16 parameters)
            try:
                out = subprocess.run([tool, raw], capture_output=True, text=True, timeout=10).stdout.strip()
            except (OSError, subprocess.SubprocessError):
                out = ""
            if out and not out.startswith("_Z"):
                name = out
                break
        else:                                                # no demangler: the length-prefixed identifiers of the nested name
            m = re.match(r"_ZN((?:\d+[A-Za-z_]\w*?)+)E", raw)
            parts, rest = [], m.group(1) if m else ""
            while rest:
                n = re.match(r"\d+", rest)
                if not n:
                    break
                k = int(n.group())
                parts.append(rest[n.end():n.end() + k])
                rest = rest[n.end() + k:]
            name = "::".join(q for q in parts if q != "_GLOBAL__N_1") or raw
    name = name.replace("void ", "").replace("(anonymous namespace)::", "")
    depth, cut = 0, len(name)
    for i, ch in enumerate(name):                           # the parameter list opens at the first '(' outside <...>
        if ch == "<":
            depth += 1
        elif ch == ">":
            depth -= 1
        elif ch == "(" and depth == 0:
            cut = i
            break
    _demangled[raw] = name[:cut].strip()
    return _demangled[raw]
