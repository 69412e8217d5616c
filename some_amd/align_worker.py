"""Worker process for the CSV word alignment (``batch_logic.align_job``).

``python -m some_amd.align_worker`` reads length-prefixed pickled jobs ``(key, args)`` from stdin and writes
``(key, result | exception)`` back on stdout.  Started by ``AlignPool`` as plain subprocesses, so a worker imports
numpy + ``some_amd.batch_logic`` only (0.2 s) - not the parent's ``__main__`` with torch and the GPU runtime, which is what
``multiprocessing``'s spawn / forkserver children would do."""
import pickle
import struct
import subprocess
import sys
import threading
from typing import Dict, List


def _read(stream):
    head = stream.read(8)
    if len(head) < 8:
        return None
    (n,) = struct.unpack('<Q', head)
    return pickle.loads(stream.read(n))


def _write(stream, obj):
    blob = pickle.dumps(obj, protocol=pickle.HIGHEST_PROTOCOL)
    stream.write(struct.pack('<Q', len(blob)))
    stream.write(blob)
    stream.flush()


def main():
    from some_amd import batch_logic
    inp, out = sys.stdin.buffer, sys.stdout.buffer
    while True:
        job = _read(inp)
        if job is None:
            return
        key, args = job
        try:
            _write(out, (key, batch_logic.align_job(*args)))
        except BaseException as e:  # noqa: BLE001
            _write(out, (key, e))


class AlignPool:
    """N worker subprocesses, jobs dealt round-robin, one reader thread per worker collecting ``results[key]``."""

    def __init__(self, workers: int):
        import os
        import pathlib
        root = str(pathlib.Path(__file__).resolve().parents[1])
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get('PYTHONPATH', ''))
        self.procs: List[subprocess.Popen] = [
            subprocess.Popen([sys.executable, '-m', 'some_amd.align_worker'], stdin=subprocess.PIPE, stdout=subprocess.PIPE, env=env)
            for _ in range(workers)]
        self.results: Dict[object, object] = {}
        self._sent = [0] * workers
        self._next = 0
        self._lock = threading.Lock()
        self._readers = [threading.Thread(target=self._drain, args=(i,), daemon=True) for i in range(workers)]
        for t in self._readers:
            t.start()

    def _drain(self, i: int):
        out = self.procs[i].stdout
        while True:
            item = _read(out)
            if item is None:
                return
            with self._lock:
                self.results[item[0]] = item[1]

    def submit(self, key, args):
        i = self._next
        self._next = (i + 1) % len(self.procs)
        self._sent[i] += 1
        _write(self.procs[i].stdin, (key, args))

    def close(self) -> Dict[object, object]:
        """Finish all jobs, stop the workers, return {key: result}; re-raises the first worker exception."""
        for p in self.procs:
            p.stdin.close()
        for t in self._readers:
            t.join()
        for p in self.procs:
            p.wait()
        if len(self.results) != sum(self._sent):
            raise RuntimeError(f'alignment workers returned {len(self.results)} of {sum(self._sent)} rows')
        for v in self.results.values():
            if isinstance(v, BaseException):
                raise v
        return self.results


if __name__ == '__main__':
    main()
