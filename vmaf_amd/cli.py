"""run_vmaf / run_vmaf_in_batch for the MI355X engine (Python 3).

Same argv contract, exit codes and text / xml / json output schema as the reference's Python-2 launchers
(run_vmaf:24-140, run_vmaf_in_batch:24-154, python/src/vmaf/core/result.py:55-98,156-293); the
VmafQualityRunner -> `feature/vmaf all` subprocess -> libsvm chain behind them
(python/src/vmaf/core/quality_runner.py:295-323) is replaced by vmaf_amd.Engine + predict_from_atoms.

Differences a user can see: content_id/asset_id of run_vmaf are a CRC of the file name (the reference
uses Python 2's hash()), "workdir" is empty (no work files are made), --local-explain is not offered.
"""
import json
import os
import re
import sys
import zlib
from collections import OrderedDict

import numpy as np

FMTS = ['yuv420p', 'yuv422p', 'yuv444p', 'yuv420p10le', 'yuv422p10le', 'yuv444p10le']
OUT_FMTS = ['text (default)', 'xml', 'json']
POOL_METHODS = ['mean', 'harmonic_mean', 'min', 'median', 'perc5', 'perc10', 'perc20']
EXECUTOR_ID = "VMAF_VF0.2.4b-0.6.1"          # mixin.py:62-64 + quality_runner.py:260 / feature_extractor.py VERSION


def _opt(argv, begin, name):
    for i in range(begin, len(argv) - 1):
        if argv[i] == name:
            return argv[i + 1]
    return None


def _flag(argv, begin, name):
    return name in argv[begin:]


def _stem(path):
    return os.path.splitext(os.path.basename(path))[0]


# ---------------------------------------------------------------------------------- pooling
def harmonic_mean(v):
    return 1.0 / np.mean(1.0 / (np.array(v) + 1.0)) - 1.0


def perc5(v):
    return np.percentile(v, 5)


def perc10(v):
    return np.percentile(v, 10)


def perc20(v):
    return np.percentile(v, 20)


POOL_FUNCS = {None: np.mean, 'mean': np.mean, 'harmonic_mean': harmonic_mean, 'min': np.min, 'median': np.median,
              'perc5': perc5, 'perc10': perc10, 'perc20': perc20}
POOL_NAMES = {np.mean: 'mean', np.min: 'amin', np.median: 'median'}      # numpy's __name__ of the era ('amin' for np.min)


def py2_str_float(v):
    """str(float) of Python 2 (and of numpy.float64 then): 12 significant digits, '.0' kept on whole numbers."""
    s = '%.12g' % float(v)
    if not any(c in s for c in '.en'):                # no point, exponent, inf or nan
        s += '.0'
    return s


def py2_repr_float64(v):
    """repr(numpy.float64) of the reference's era, which is what its json.dumps printed aggregates with."""
    s = '%.17g' % float(v)
    if not any(c in s for c in '.en'):
        s += '.0'
    return s


class Asset:
    def __init__(self, dataset, content_id, asset_id, ref_path, dis_path, width, height, yuv_type):
        self.dataset, self.content_id, self.asset_id = dataset, content_id, asset_id
        self.ref_path, self.dis_path = ref_path, dis_path
        self.asset_dict = {'width': width, 'height': height, 'yuv_type': yuv_type}

    def __str__(self):                           # asset.py:344-440
        w, h, t = self.asset_dict['width'], self.asset_dict['height'], self.asset_dict['yuv_type']
        suffix = "" if t == 'yuv420p' else "_" + t
        return (f"{self.dataset}_{self.content_id}_{self.asset_id}_{_stem(self.ref_path)}_{w}x{h}{suffix}_vs_"
                f"{_stem(self.dis_path)}_{w}x{h}{suffix}_q_{w}x{h}")

    def to_full_repr(self):                      # asset.py:473-474: json of __dict__, keys sorted
        d = {'asset_dict': self.asset_dict, 'asset_id': self.asset_id, 'content_id': self.content_id,
             'dataset': self.dataset, 'dis_path': self.dis_path, 'ref_path': self.ref_path, 'workdir': ""}
        return json.dumps(d, sort_keys=True)


class Result:
    """result.py:55-293 for the VMAF runner: per-frame lists keyed '<X>_scores', aggregates '<X>_score'."""

    def __init__(self, asset, scores, executor_id=EXECUTOR_ID):
        self.asset, self.scores = asset, scores                      # scores: dict '<X>_scores' -> list
        self.executor_id = executor_id
        self.method = np.mean

    def set_pool(self, name):
        self.method = POOL_FUNCS[name]

    @property
    def method_name(self):
        return POOL_NAMES.get(self.method, getattr(self.method, '__name__', 'mean'))

    def keys(self):
        return sorted(self.scores)

    def aggregate(self, scores_key):
        return float(self.method(self.scores[scores_key]))

    def to_string(self):
        ks = self.keys()
        s = f"Asset: {self.asset.to_full_repr()}\nExecutor: {self.executor_id}\nResult:\n"
        n = len(self.scores[ks[0]])
        for i in range(n):
            s += f"Frame {i}: " + ", ".join(f"{k[:-1]}:{self.scores[k][i]:.6f}" for k in ks) + "\n"
        s += f"Aggregate ({self.method_name}): " + ", ".join(f"{k[:-1]}:{self.aggregate(k):.6f}" for k in ks)
        return s

    def to_dict(self):
        ks = self.keys()
        top = OrderedDict()
        top['executorId'] = self.executor_id
        top['asset'] = {'identifier': str(self.asset)}
        top['frames'] = []
        for i in range(len(self.scores[ks[0]])):
            fr = OrderedDict(frameNum=i)
            for k in ks:
                fr[k[:-1]] = float(self.scores[k][i])
            top['frames'].append(fr)
        top['aggregate'] = OrderedDict((k[:-1], self.aggregate(k)) for k in ks)
        top['aggregate']['method'] = self.method_name
        return top

    def to_json(self):
        """result.py:289-293.  Python 2's json printed every float with repr(): the per-frame scores (Python floats) in
        their shortest form, the aggregates (numpy.float64 from the pooling function) with numpy's 17 significant digits
        of the era - "VMAF_score": 65.44885887590759 per frame, "VMAF_score": 0.93335333333333337 in the aggregate
        (README.md:131-140, python/test/result_test.py:192-197).  Both parse to the same doubles."""
        d = self.to_dict()
        marks = {}
        for k, v in list(d['aggregate'].items()):
            if isinstance(v, float):
                mark = "@@AGG%d@@" % len(marks)
                marks['"%s"' % mark] = py2_repr_float64(v)
                d['aggregate'][k] = mark
        text = json.dumps(d, sort_keys=False, indent=4, separators=(', ', ': '))   # Python 2.7 kept ', ' before the line break
        for mark, num in marks.items():
            text = text.replace(mark, num)
        return text

    def to_xml(self):
        """result.py:178-224 as Python 2 printed it: values through str() (12 significant digits), attributes in plain
        string order, which puts frameNum and method last (python/test/result_test.py:148-160 pins both)."""
        from xml.dom import minidom
        from xml.etree import ElementTree
        ks = self.keys()

        def set_sorted(elem, attrs):
            for name in sorted(attrs):
                elem.set(name, attrs[name])

        top = ElementTree.Element('result')
        top.set('executorId', self.executor_id)
        ElementTree.SubElement(top, 'asset').set('identifier', str(self.asset))
        frames = ElementTree.SubElement(top, 'frames')
        for i in range(len(self.scores[ks[0]])):
            attrs = {k[:-1]: py2_str_float(self.scores[k][i]) for k in ks}
            attrs['frameNum'] = str(i)
            set_sorted(ElementTree.SubElement(frames, 'frame'), attrs)
        attrs = {k[:-1]: py2_str_float(self.aggregate(k)) for k in ks}
        attrs['method'] = self.method_name
        set_sorted(ElementTree.SubElement(top, 'aggregate'), attrs)
        return minidom.parseString(ElementTree.tostring(top, 'utf-8')).toprettyxml(indent="  ")


# ---------------------------------------------------------------------------------- scoring
class LumaFile:
    """A raw YUV file seen as its luma planes, without reading it: a memory map plus the frame geometry.  The reference
    reads one frame at a time (feature/src/common/frame.c:26-129: Y plane, then fread-skips the chroma); here a batch of
    planes is copied out of the map when the GPU is ready for it, so a clip of any length is scored in constant memory."""

    def __init__(self, path, fmt, width, height):
        self.ten = fmt.endswith('10le')
        self.w, self.h = width, height
        n = width * height
        chroma = {'yuv420p': n // 2, 'yuv422p': n, 'yuv444p': 2 * n}[fmt.replace('10le', '')]
        self.dtype = np.dtype('<u2') if self.ten else np.dtype(np.uint8)
        self.n, self.frame_samples = n, n + chroma
        total = os.path.getsize(path) // self.dtype.itemsize
        self.frames = total // self.frame_samples
        if total - self.frames * self.frame_samples >= n:
            self.frames += 1                               # a last frame with short chroma is still scored (frame.c:83-87)
        self.mm = np.memmap(path, dtype=self.dtype, mode='r') if total else np.zeros(0, self.dtype)
        self.fd = os.open(path, os.O_RDONLY) if total else -1

    def __del__(self):
        if getattr(self, 'fd', -1) >= 0:
            os.close(self.fd)
            self.fd = -1

    def __len__(self):
        return self.frames

    def planes(self, start, stop, out=None):
        """Raw luma samples of frames [start, stop) as [n,h,w] (uint8 / uint16), copied into `out[:n]` when given."""
        n = stop - start
        if out is None:
            out = np.empty((n, self.h, self.w), self.dtype)
        if self.fd >= 0 and out.flags.c_contiguous:
            # positional reads straight into the (page-locked) destination: no page faults on a mapping, no GIL while the
            # kernel copies, and a frame counts when its luma is complete (frame.c:83-87), so every read is a full plane
            size, step = self.n * self.dtype.itemsize, self.frame_samples * self.dtype.itemsize
            for f in range(start, stop):
                buf = memoryview(out[f - start]).cast('B')
                got = 0
                while got < size:
                    k = os.preadv(self.fd, [buf[got:]], f * step + got)
                    if k <= 0:
                        raise IOError('short read in frame %d' % f)
                    got += k
            return out[:n]
        whole = min(stop, len(self.mm) // self.frame_samples)        # frames whose chroma is complete too
        if whole > start:
            v = self.mm[start * self.frame_samples:whole * self.frame_samples].reshape(whole - start, self.frame_samples)
            np.copyto(out[:whole - start].reshape(whole - start, self.n), v[:, :self.n])
        for f in range(max(whole, start), stop):                     # the truncated last frame
            out[f - start] = np.asarray(self.mm[f * self.frame_samples:f * self.frame_samples + self.n]).reshape(self.h, self.w)
        return out[:n]


def read_luma(path, fmt, width, height, raw=False):
    """The luma planes [F,h,w] of a whole raw YUV file, chroma skipped (frame.c:80-124): as float32 in 8-bit units
    (feature/src/common/file_io.c:86-172: 10-bit / 4.0), or with raw=True as the uint8 / uint16 samples themselves, for
    the GPU to widen.  For small files (tests, batch-mode pairs); the scorer streams through LumaFile instead."""
    lf = LumaFile(path, fmt, width, height)
    y = lf.planes(0, len(lf))
    if raw:
        return y
    return (y.astype(np.float64) / 4.0).astype(np.float32) if lf.ten else y.astype(np.float32)


def load_asset(asset):
    """-> (ref, dis) raw luma [F,h,w] of one asset (host side only; safe to call from reader threads)."""
    w, h, fmt = asset.asset_dict['width'], asset.asset_dict['height'], asset.asset_dict['yuv_type']
    return read_luma(asset.ref_path, fmt, w, h, raw=True), read_luma(asset.dis_path, fmt, w, h, raw=True)


PREFETCH_MAX_PAIR_BYTES = 1 << 30          # pairs larger than this are not loaded ahead: the scorer streams them from their files


def prefetched_assets(assets, threads=4, max_bytes=2 << 30):
    """Yield (asset, luma) in order while reader threads load the files of the next assets (up to max_bytes of input
    ahead): a single thread copies ~8 GB/s out of the page cache, which is what bounds batch mode otherwise.  luma =
    (ref, dis) arrays, or None for a pair too large to hold (score_asset then streams it batch by batch)."""
    from concurrent.futures import ThreadPoolExecutor

    def size(a):
        try:
            return os.path.getsize(a.ref_path) + os.path.getsize(a.dis_path)
        except OSError:
            return 0

    if len(assets) <= 1 or threads <= 1:
        for a in assets:
            yield a, (load_asset(a) if size(a) <= PREFETCH_MAX_PAIR_BYTES else None)
        return
    with ThreadPoolExecutor(max_workers=threads) as pool:
        pending, ahead, nxt = [], 0, 0
        while nxt < len(assets) or pending:
            while nxt < len(assets) and (not pending or (len(pending) < 2 * threads and ahead + size(assets[nxt]) <= max_bytes)):
                a = assets[nxt]
                w, h, fmt = a.asset_dict['width'], a.asset_dict['height'], a.asset_dict['yuv_type']
                if size(a) > PREFETCH_MAX_PAIR_BYTES:
                    pending.append((a, 0, None, None))
                else:
                    pending.append((a, size(a), pool.submit(read_luma, a.ref_path, fmt, w, h, True),
                                    pool.submit(read_luma, a.dis_path, fmt, w, h, True)))
                ahead += pending[-1][1]
                nxt += 1
            a, sz, fr, fd = pending.pop(0)
            ahead -= sz
            yield a, ((fr.result(), fd.result()) if fr is not None else None)


def _batch_size(w, h):
    return int(min(64, max(4, (128 << 20) // (w * h * 4))))


def _read_threads(shards=1):
    """Reader threads PER SHARD, the rule of vmafossexec (libvmaf.cpp fetch_threads): VMAF_HIP_READ_THREADS if set, else
    min(8, CPUs of this process / shards) but never fewer than 4 - one copying thread moves ~8 GB/s out of the page cache,
    a shard's PCIe link ~51 GB/s, so eight shards with one reader each would together run at the rate of ONE well-fed GPU."""
    try:
        n = int(os.environ.get("VMAF_HIP_READ_THREADS", "0"))
    except ValueError:
        n = 0
    if n > 0:
        return min(n, 64)
    try:
        cpus = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cpus = os.cpu_count() or 8
    return max(4, min(8, cpus // max(1, shards)))


def _stream_range(eng, ref, dis, start, stop, atoms, readers=1):
    """Frames [start, stop) of one pair through one engine, streamed: two page-locked batch buffers per file, the copy out
    of the file map for batch k+1 overlaps the H2D copies and kernels of batch k (two calls in flight).  ref / dis are
    LumaFile objects or in-memory arrays [F,h,w].  The shard seeds its own motion halo (ref frame start-1).  The planes of
    a batch are copied by `readers` threads side by side (numpy releases the GIL while it copies): one thread moves ~8 GB/s
    out of the page cache, less than half of what the PCIe link takes."""
    from concurrent.futures import ThreadPoolExecutor
    from . import binding
    if stop <= start:
        return
    B = eng.max_batch
    get = (lambda src, a, b, out: src.planes(a, b, out)) if isinstance(ref, LumaFile) else \
          (lambda src, a, b, out: (np.copyto(out[:b - a], src[a:b]), out[:b - a])[1])
    dt = ref.dtype
    pins = [(binding.PinnedArray((B, eng.h, eng.w), dt), binding.PinnedArray((B, eng.h, eng.w), dt)) for _ in range(2)]
    pool = ThreadPoolExecutor(max_workers=readers) if readers > 1 else None

    def fill(at, n, pr, pd):
        if pool is None:
            return get(ref, at, at + n, pr.array), get(dis, at, at + n, pd.array)
        per = max(1, (2 * n + readers - 1) // readers)                  # frames per task, over both files
        tasks = [pool.submit(get, src, at + o, at + min(n, o + per), pin.array[o:])
                 for src, pin in ((ref, pr), (dis, pd)) for o in range(0, n, per)]
        for t in tasks:
            t.result()
        return pr.array[:n], pd.array[:n]

    try:
        if start > 0:
            eng.motion_seed(get(ref, start - 1, start, pins[0][0].array)[0])
        inflight, k = 0, 0
        for at in range(start, stop, B):
            n = min(B, stop - at)
            if inflight == 2:
                eng.sync_oldest()
                inflight -= 1
            r, d = fill(at, n, *pins[k & 1])
            eng.extract_host_async(r, d, atoms[at:at + n], continue_sequence=at > 0)
            inflight += 1
            k += 1
        eng.sync()
    finally:
        eng.sync()
        if pool is not None:
            pool.shutdown()
        for pr, pd in pins:
            pr.close()
            pd.close()


def score_asset(asset, model_path, phone_model, device=0, engine_cache=None, luma=None, devices=None):
    """-> Result.  The feature + fusion hot path on the GPU; no CPU fallback.  The pair is streamed from its files (or
    taken from `luma` = (ref, dis) arrays a prefetcher already holds).  devices = [a, b, ..]: contiguous frame ranges of
    this ONE pair on several GPUs, one host thread + engine per entry (an id may repeat), each with its one-frame motion
    halo; motion2 and the fusion run once on the merged atoms."""
    from . import binding, sharding
    w, h, fmt = asset.asset_dict['width'], asset.asset_dict['height'], asset.asset_dict['yuv_type']
    if luma is not None:
        ref, dis = luma
    else:
        ref, dis = LumaFile(asset.ref_path, fmt, w, h), LumaFile(asset.dis_path, fmt, w, h)
    frames = min(len(ref), len(dis))
    if frames <= 0:
        raise binding.VmafHipError(f"no complete frame in {asset.ref_path} / {asset.dis_path}")
    devices = list(devices) if devices else [device]
    devices = devices[:frames]
    atoms = np.zeros((frames, binding.N_ATOMS))

    def engine_for(dev, slot):
        key = (w, h, dev, slot)
        eng = engine_cache.get(key) if engine_cache is not None else None
        if eng is None:
            eng = binding.Engine(w, h, max_batch=_batch_size(w, h), device=dev)
            if engine_cache is not None:
                engine_cache[key] = eng
        return eng

    if len(devices) == 1:
        eng = engine_for(devices[0], 0)
        try:
            _stream_range(eng, ref, dis, 0, frames, atoms, readers=_read_threads(1))
        finally:
            if engine_cache is None:
                eng.close()
    else:
        import threading
        errs = []

        def work(g):
            try:
                eng = engine_for(devices[g], g)
                try:
                    _stream_range(eng, ref, dis, *sharding.frame_range(frames, len(devices), g), atoms, readers=_read_threads(len(devices)))
                finally:
                    if engine_cache is None:
                        eng.close()
            except BaseException as ex:                           # noqa: BLE001 - re-raised below on the calling thread
                errs.append(ex)

        th = [threading.Thread(target=work, args=(g,)) for g in range(len(devices))]
        for t in th:
            t.start()
        for t in th:
            t.join()
        if errs:
            raise errs[0]
    model_path = model_path or binding.DEFAULT_MODEL
    names = binding.model_feature_names(model_path)
    vm, feats = binding.predict_from_atoms(atoms, model_path, enable_transform=phone_model, device=devices[0])
    scores = {name + 's': feats[:, j].tolist() for j, name in enumerate(names)}
    scores['VMAF_scores'] = vm.tolist()
    return Result(asset, scores)


# ---------------------------------------------------------------------------------- run_vmaf
def _usage_run_vmaf(prog):
    print("usage: " + prog + " fmt width height ref_path dis_path [--model model_path] [--out-fmt out_fmt] [--phone-model]\n")
    print("fmt:\n\t" + "\n\t".join(FMTS) + "\n")
    print("out_fmt:\n\t" + "\n\t".join(OUT_FMTS) + "\n")


def _print_result(result, out_fmt):
    if out_fmt == 'xml':
        print(result.to_xml())
    elif out_fmt == 'json':
        print(result.to_json())
    else:
        print(result.to_string())


def run_vmaf_main(argv):
    prog = os.path.basename(argv[0])
    if len(argv) < 6:
        _usage_run_vmaf(prog)
        return 2
    try:
        fmt, width, height, ref_file, dis_file = argv[1], int(argv[2]), int(argv[3]), argv[4], argv[5]
    except ValueError:
        _usage_run_vmaf(prog)
        return 2
    if width < 0 or height < 0:
        print(f"width and height must be non-negative, but are {width} and {height}")
        _usage_run_vmaf(prog)
        return 2
    if fmt not in FMTS:
        _usage_run_vmaf(prog)
        return 2
    model_path = _opt(argv, 6, '--model')
    out_fmt = _opt(argv, 6, '--out-fmt')
    if out_fmt not in (None, 'xml', 'json', 'text'):
        _usage_run_vmaf(prog)
        return 2
    pool_method = _opt(argv, 6, '--pool')
    if pool_method not in (None,) + tuple(POOL_METHODS):
        print('--pool can only have option among {}'.format(', '.join(POOL_METHODS)))
        return 2
    phone_model = _flag(argv, 6, '--phone-model')
    cid = zlib.crc32(_stem(ref_file).encode()) % (10 ** 16)
    asset = Asset("cmd", cid, cid, ref_file, dis_file, width, height, fmt)
    # VMAF_HIP_DEVICES="0,1,..": contiguous frame ranges of this one pair on several GPUs (default: device 0)
    result = score_asset(asset, model_path, phone_model, devices=_device_list() if os.environ.get("VMAF_HIP_DEVICES", "").strip() else [0])
    result.set_pool(pool_method)
    _print_result(result, out_fmt)
    return 0


# ---------------------------------------------------------------------------------- run_vmaf_in_batch
def _usage_batch(prog):
    print("usage: " + prog + " input_file [--model model_path] [--out-fmt out_fmt] [--parallelize] [--phone-model]\n")
    print("out_fmt:\n\t" + "\n\t".join(OUT_FMTS) + "\n")
    print("input_file contains lines of:")
    print("\tfmt width height ref_path dis_path\\n")
    print("fmt:\n\t" + "\n\t".join(FMTS) + "\n")


def parse_batch_file(path, prog="run_vmaf_in_batch"):
    """-> (assets, 0) or (None, 1) on a bad line (run_vmaf_in_batch:60-96)."""
    assets = []
    with open(path, "rt") as f:
        for line in f.readlines():
            if re.match(r"^#", line):
                print("Skip commented line: {}".format(line))
                continue
            if re.match(r"[\s]+", line):
                continue
            mo = re.match(r"([\S]+) ([0-9]+) ([0-9]+) ([\S]+) ([\S]+)", line)
            if not mo or mo.group(1) not in FMTS:
                print("Unknown format: {}".format(line))
                _usage_batch(prog)
                return None, 1
            assets.append(Asset("cmd", 0, len(assets), mo.group(4), mo.group(5), int(mo.group(2)), int(mo.group(3)), mo.group(1)))
    return assets, 0


def _device_list():
    """GPUs to spread work over: VMAF_HIP_DEVICES="a,b,.." when set (the knob compute_vmaf and vmafossexec read too),
    else every visible device."""
    from . import binding
    env = os.environ.get("VMAF_HIP_DEVICES", "").strip()
    if env:
        try:
            devs = [int(t) for t in env.split(",") if t.strip() != ""]
        except ValueError:
            raise binding.VmafHipError(f"VMAF_HIP_DEVICES={env!r} is not a comma-separated list of device numbers")
        if devs:
            return devs
    return list(range(max(1, binding.device_count())))


def _batch_worker(device, items, model_path, phone_model, q):
    """One worker process per GPU.  Posts (index, scores) per asset, ("error", index, message) when an asset fails, and
    always the None sentinel last - the parent also watches the process itself, so a worker that dies without a word
    (a signal, the OOM killer) still ends the run with an error instead of a hang."""
    current = items[0][0] if items else -1
    try:
        cache = {}
        index = {id(asset): idx for idx, asset in items}
        for asset, luma in prefetched_assets([a for _, a in items]):
            current = index[id(asset)]
            q.put((current, score_asset(asset, model_path, phone_model, device=device, engine_cache=cache, luma=luma).scores))
    except BaseException as ex:                                   # noqa: BLE001 - reported to the parent, which exits non-zero
        q.put(("error", current, f"{type(ex).__name__}: {ex}"))
    finally:
        q.put(None)


def _collect_from_workers(procs, q, assets, results, poll_s=0.5):
    """Drain the workers' queue until every worker has posted its sentinel.  -> None, or the message of the first failure
    (an error item, or a worker process that ended without its sentinel).  Never waits on a dead worker."""
    import queue as queue_mod
    done, failure = 0, None
    while done < len(procs) and failure is None:
        try:
            item = q.get(timeout=poll_s)
        except queue_mod.Empty:
            # the queue is empty: a worker that is no longer alive will never post again
            dead = [p for p in procs if not p.is_alive()]
            if len(dead) > done:
                try:
                    item = q.get(timeout=poll_s)                  # items still in the pipe of a worker that just finished
                except queue_mod.Empty:
                    p = dead[0]
                    failure = f"worker process {p.pid} ended with exit code {p.exitcode} before finishing its assets"
                    break
            else:
                continue
        if item is None:
            done += 1
        elif item[0] == "error":
            a = assets[item[1]] if 0 <= item[1] < len(assets) else None
            failure = f"asset {item[1]} ({a.ref_path} vs {a.dis_path}): {item[2]}" if a else item[2]
        else:
            results[item[0]] = Result(assets[item[0]], item[1])
    if failure is not None:
        for p in procs:                                           # exact processes this function's caller started
            if p.is_alive():
                p.terminate()
    for p in procs:
        p.join(timeout=30)
    if failure is None:
        bad = [p for p in procs if p.exitcode not in (0, None)]
        if bad:
            failure = f"worker process {bad[0].pid} ended with exit code {bad[0].exitcode}"
    return failure


def run_vmaf_in_batch_main(argv):
    prog = os.path.basename(argv[0])
    if len(argv) < 2:
        _usage_batch(prog)
        return 2
    model_path = _opt(argv, 2, '--model')
    out_fmt = _opt(argv, 2, '--out-fmt')
    if out_fmt not in (None, 'xml', 'json', 'text'):
        _usage_batch(prog)
        return 2
    pool_method = _opt(argv, 2, '--pool')
    if pool_method not in (None,) + tuple(POOL_METHODS):
        print('--pool can only have option among {}'.format(', '.join(POOL_METHODS)))
        return 2
    parallelize = _flag(argv, 2, '--parallelize')
    phone_model = _flag(argv, 2, '--phone-model')
    assets, rc = parse_batch_file(argv[1], prog)
    if rc:
        return rc

    results = [None] * len(assets)
    from . import binding, sharding
    # --parallelize: one worker process per GPU, or per entry of VMAF_HIP_DEVICES ("0,1,2,3"; an id may repeat, which puts
    # several workers on one GPU - that is how the worker path is exercised on a one-GPU box)
    devices = _device_list() if parallelize else [0]
    ndev = len(devices)
    if parallelize and ndev > 1 and len(assets) > 1:
        # whole pairs per GPU, longest first, one worker process per GPU; results back in input order
        # (the reference forks one process per asset, executor.py:100-123 - here one per device)
        import multiprocessing as mp
        costs = [a.asset_dict['width'] * a.asset_dict['height'] * max(1, os.path.getsize(a.ref_path)) for a in assets]
        parts = sharding.assign_pairs(costs, ndev)
        if os.environ.get("VMAF_HIP_TRACE", "0") not in ("", "0"):
            for d in range(ndev):                              # who scores what (tests check the balance through this)
                print("run_vmaf_in_batch: worker %d (device %d): assets %s, cost %d" % (d, devices[d], parts[d], sum(costs[i] for i in parts[d])),
                      file=sys.stderr)
        ctx = mp.get_context("spawn")
        q = ctx.Queue()
        procs = [ctx.Process(target=_batch_worker, args=(devices[d], [(i, assets[i]) for i in parts[d]], model_path, phone_model, q))
                 for d in range(ndev) if parts[d]]
        for p in procs:
            p.start()
        failure = _collect_from_workers(procs, q, assets, results)
        if failure is None and any(r is None for r in results):
            failure = "no result for asset %d" % [r is None for r in results].index(True)
        if failure is not None:
            print("run_vmaf_in_batch: " + failure, file=sys.stderr)
            return 1
    else:
        cache = {}
        for i, (a, luma) in enumerate(prefetched_assets(assets)):
            results[i] = score_asset(a, model_path, phone_model, engine_cache=cache, luma=luma)

    for result in results:
        result.set_pool(pool_method)
        if out_fmt == 'xml':
            print(result.to_xml())
        elif out_fmt == 'json':
            print(result.to_json())
        else:
            print('============================')
            print('Asset {asset_id}:'.format(asset_id=result.asset.asset_id))
            print('============================')
            print(result.to_string())
    return 0


if __name__ == "__main__":
    sys.exit(run_vmaf_main(sys.argv))
