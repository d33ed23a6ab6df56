"""Per-launch census of the operator library, taken AT the C ABI (bench.py's roofline leg and the developer shape view).

`with Census(lib) as c:` replaces every compute entry point of the loaded ctypes library by a wrapper that records a HIP
event on the launch stream before and after the call and derives (family, algorithmic flops, algorithmic bytes, shape tag)
from the C arguments themselves; `c.records` is the list.  Nothing in stabletriton_amd/ knows about it: the product
launchers (ops.py) look their entry points up on the library object at every call, which is what gets wrapped here.
"""
import torch

GEGLU = 4
ES = {0: 4, 1: 2, 2: 2, 3: 4}    # ST_F32, ST_BF16, ST_F16, ST_F32S (split fp32: 4 bytes per value)


def _linear(a):
    M, N, K, epi, dt = a[6], a[7], a[8], a[13], a[14]
    rows = 2 * N if epi & GEGLU else N
    return "linear", 2.0 * M * rows * K, float((M * K + rows * K + M * N) * ES[dt]), f"M={M} N={N} K={K} epi={epi}" + (" stats" if a[17] else "") + (" colstats" if a[20] else "")


def _ln_linear(a):
    M, N, K, epi, dt = a[7], a[8], a[9], a[13], a[14]
    rows = 2 * N if epi & GEGLU else N
    return "linear", 2.0 * M * rows * K, float((M * K + rows * K + M * N) * ES[dt]), f"M={M} N={N} K={K} ln" + (" geglu" if epi & GEGLU else "")


def _ln_linear_xattn(a):
    M, N, K, rpb, S = a[9], a[10], a[11], a[15], a[16]
    es = ES[a[21]]
    return ("linear_xattn", 2.0 * M * N * K + 4.0 * M * S * N, float((M * K + N * K + M * N + 2 * (M // rpb) * S * N) * es),
            f"M={M} N={N} K={K} ln xattn S={S}")


def _attention(a):
    B, T, S, H, D, dt = a[4], a[5], a[6], a[7], a[8], a[14]
    return ("attention_self" if S == T else "attention_cross", 4.0 * B * H * T * S * D, float((2 * B * T + 2 * B * S) * H * D * ES[dt]),
            f"B={B} T={T} S={S} H={H}")


def _attention_split(a):
    B, T, S, H, D = a[4], a[5], a[6], a[7], a[8]
    return ("attention_self" if S == T else "attention_cross", 4.0 * B * H * T * S * D, float((2 * B * T + 2 * B * S) * H * D * 4),
            f"B={B} T={T} S={S} H={H} k/v split images")


def _conv2d(a):
    N, Hin, Win, Cin, Cout, R, S, stride, pad, ups, epi, dt = a[6:18]
    He, We = (2 * Hin, 2 * Win) if ups else (Hin, Win)
    Ho, Wo = (He + 2 * pad - R) // stride + 1, (We + 2 * pad - S) // stride + 1
    return ("conv2d", 2.0 * N * Ho * Wo * Cout * R * S * Cin, float((N * Hin * Win * Cin + Cout * R * S * Cin + N * Ho * Wo * Cout) * ES[dt]),
            f"Cin={Cin} H={Hin} Cout={Cout} k={R} s={stride} ups={ups} epi={epi}" + (" colstats" if a[20] else ""))


def _group_norm(a):
    N, C, HW, silu, dt = a[4], a[5], a[6], a[9], a[11]
    return "group_norm", 0.0, 2.0 * N * C * HW * ES[dt], f"N={N} C={C} HW={HW} silu={silu}"


def _group_norm_from_stats(a):
    N, C, HW, silu, dt = a[4], a[5], a[6], a[9], a[10]
    return "group_norm", 0.0, 2.0 * N * C * HW * ES[dt], f"N={N} C={C} HW={HW} silu={silu} from-stats"


def _group_norm_from_stats_cat(a):
    N, C, HW, silu, dt = a[5], a[6], a[7], a[10], a[11]
    return "group_norm", 0.0, 2.0 * N * C * HW * ES[dt], f"N={N} C={C} HW={HW} silu={silu} from-stats, two sources"


def _conv1x1_cat(a):
    C0, C1, N, H, W, Cout, epi, dt = a[1], a[3], a[8], a[9], a[10], a[11], a[12], a[13]
    Cin = C0 + C1
    return ("conv2d", 2.0 * N * H * W * Cout * Cin, float((N * H * W * Cin + Cout * Cin + N * H * W * Cout) * ES[dt]),
            f"Cin={C0}+{C1} H={H} Cout={Cout} k=1 s=1 ups=0 epi={epi}" + (" colstats" if a[16] else ""))


def _layer_norm(a):
    rows, C, dt = a[4], a[5], a[7]
    return "layer_norm", 0.0, 2.0 * rows * C * ES[dt], f"rows={rows} C={C}"


def _geglu(a):
    rows, F, dt = a[3], a[4], a[8]
    return "geglu", 0.0, 3.0 * rows * F * ES[dt], f"rows={rows} F={F}"


def _quantize_fp8(a):
    rows, C, dt = a[4], a[5], a[6]
    return "quantize_fp8", 0.0, float(rows * C * (ES[dt] + 1)), f"rows={rows} C={C}"


def _ln_quantize_fp8(a):
    rows, C, dt = a[5], a[6], a[8]
    return "quantize_fp8", 0.0, float(rows * C * (ES[dt] + 1)), f"rows={rows} C={C} ln"


def _linear_fp8(a):
    M, N, K, epi = a[7], a[8], a[9], a[13]
    rows = 2 * N if epi & GEGLU else N
    return "linear_fp8", 2.0 * M * rows * K, float(M * K + rows * K + 2 * M * N), f"M={M} N={N} K={K} fp8 epi={epi}"


def _linear_emit8(a):
    fam, fl, by, tag = _linear(a)
    return fam, fl, by + float(a[6] * a[7]), tag + " +e4m3 copy"


def _linear_fp8x(a):
    M, N, K, epi = a[8], a[9], a[10], a[14]
    rows = 2 * N if epi & GEGLU else N
    out_bytes = (2 * M * N if a[7] else 0) + (M * N if a[23] else 0)
    return "linear_fp8", 2.0 * M * rows * K, float(M * K + rows * K + out_bytes), f"M={M} N={N} K={K} fp8x epi={epi}" + (" ln" if a[17] else "") + (" +e4m3 copy" if a[23] else "")


def _split_f32(a):
    rows, K = a[2], a[3]
    return "split_f32", 0.0, 8.0 * rows * K, f"rows={rows} K={K}"


DECODERS = {"st_split_f32": _split_f32, "st_attention_split": _attention_split, "st_linear": _linear, "st_linear_emit8": _linear_emit8, "st_linear_fp8x": _linear_fp8x, "st_ln_linear": _ln_linear, "st_ln_linear_xattn": _ln_linear_xattn, "st_attention": _attention,
            "st_conv2d": _conv2d, "st_group_norm": _group_norm, "st_group_norm_from_stats": _group_norm_from_stats,
            "st_group_norm_from_stats_cat": _group_norm_from_stats_cat, "st_conv1x1_cat": _conv1x1_cat,
            "st_layer_norm": _layer_norm, "st_geglu": _geglu, "st_quantize_fp8": _quantize_fp8,
            "st_layer_norm_quantize_fp8": _ln_quantize_fp8, "st_linear_fp8": _linear_fp8}


class Census:
    def __init__(self, lib):
        self.lib = lib
        self.records = []            # (family, flops, bytes, start event, end event, shape tag)
        self._saved = {}

    def _wrap(self, name, fn, decode):
        def call(*args):
            fam, flops, nbytes, tag = decode(args)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            status = fn(*args)
            e1.record()
            self.records.append((fam, flops, nbytes, e0, e1, tag))
            return status
        return call

    def __enter__(self):
        for name, decode in DECODERS.items():
            fn = getattr(self.lib, name)
            self._saved[name] = fn
            setattr(self.lib, name, self._wrap(name, fn, decode))
        return self

    def __exit__(self, *exc):
        for name, fn in self._saved.items():
            setattr(self.lib, name, fn)
        self._saved.clear()
        return False
