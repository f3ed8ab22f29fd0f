"""Bounded failure for multi-rank runs: a stage tracker, a deadline thread and a one-line failure record.

A multi-GPU launch has code that runs for the first time on the box it is deployed on (process-group start-up, the hipIpc
handle exchange and self-validation of the one-shot all-reduce, the first RCCL send / recv of the draft hand-off -- reference
ssd/engine/llm_engine.py:61-93, ssd/engine/speculator_async.py:130-187).  Whatever goes wrong there, a launch must END, within
a bounded time, with a record that names the stage -- never hang, never die silently:

  * the engine and the benchmark announce what they are about to do with ``stage(name, timeout)``;
  * a daemon thread ends the process (``os._exit``) when the stage's deadline or the run's total deadline passes -- it runs even
    while the main thread is blocked inside a collective (those calls release the GIL);
  * SIGTERM (the launcher tearing the group down because ANOTHER rank failed) is caught through a wake-up file descriptor and
    handled by the same thread, for the same reason: a Python-level handler would wait for the blocked call to return;
  * every failing rank leaves its record in a directory shared by the launch; rank 0 prints ONE json line on stdout -- its own
    record plus the peers' -- so a driver that parses "the one JSON line" gets the failing stage instead of nothing.

Nothing here touches the data path; without ``RunGuard.install()`` every call is a no-op.
"""
from __future__ import annotations

import json
import os
import signal
import socket
import sys
import threading
import time
import traceback

_ACTIVE: "RunGuard | None" = None


def stage(name: str, timeout: float | None = None) -> None:
    """Announce the next stage of this process (no-op unless a RunGuard is installed)."""
    g = _ACTIVE
    if g is not None:
        g.stage(name, timeout)


def current_stage() -> str | None:
    return None if _ACTIVE is None else _ACTIVE.cur


def env_float(name: str, default: float) -> float:
    try:
        return float(os.environ.get(name, default))
    except ValueError:
        return default


class RunGuard:
    EXIT_EXCEPTION, EXIT_TIMEOUT, EXIT_TERMINATED = 2, 3, 4

    def __init__(self, rank: int, world: int, base_record: dict | None = None, stage_timeout: float | None = None,
                 total_deadline: float | None = None, share_dir: str | None = None, stdout_rank: int = 0):
        self.rank, self.world = rank, world
        self.base = dict(base_record or {})
        self.stage_timeout = stage_timeout if stage_timeout is not None else env_float("SSD_STAGE_TIMEOUT_S", 900.0)
        self.total_deadline = total_deadline if total_deadline is not None else env_float("SSD_TOTAL_DEADLINE_S", 3000.0)
        port = os.environ.get("MASTER_PORT", "0")
        # failure records of one launch meet in a directory only its own user can write (the launcher may name it: SSD_FAIL_DIR)
        self.share_dir = share_dir or os.environ.get("SSD_FAIL_DIR") or os.path.join("/tmp", f"ssd_run_{os.getuid()}_{port}")
        self.wall0 = time.time()
        self.stdout_rank = stdout_rank
        self.t0 = time.monotonic()
        self.cur, self.cur_t0, self.cur_deadline = "start", self.t0, self.t0 + self.stage_timeout
        self.history: list[tuple[str, float]] = []
        self._lock = threading.Lock()
        self._done = False
        self._failed = False
        self._rfd = self._wfd = None

    # ---- stage bookkeeping ----
    def stage(self, name: str, timeout: float | None = None) -> None:
        now = time.monotonic()
        with self._lock:
            self.history.append((self.cur, round(now - self.cur_t0, 3)))
            self.cur, self.cur_t0 = name, now
            self.cur_deadline = now + (timeout if timeout is not None else self.stage_timeout)

    def done(self) -> None:
        """The result line is out.  From here on SIGTERM ends the process as it would without a guard (the no-op handler that feeds the
        reader thread must not swallow it during the farewell barrier)."""
        self._done = True
        if self._wfd is not None and threading.current_thread() is threading.main_thread():
            try:
                signal.set_wakeup_fd(-1)
                signal.signal(signal.SIGTERM, signal.SIG_DFL)
            except Exception:
                pass

    # the code paths that execute for the FIRST time on a multi-GPU box, by the stage that contains them.  A path is
    # "completed" once the run has moved past its stage (paths 6, 7, 9, 10 run inside the first captures / the timed steps).
    FIRST_RUN_PATHS = (("process_group_init", 1), ("subgroup_creation", 2), ("kv_blocks_agree", 3), ("one_shot_allreduce_validation", 4),
                       ("one_shot_allreduce_ipc_exchange", 5), ("ttft (", 6), ("ttft (", 7), ("draft_hello", 8), ("timed_steps", 9),
                       ("timed_steps", 10))

    def first_run_paths(self) -> dict:
        """Which of DESIGN section 7's first-run paths this rank has been through (stage left behind), and which one it is in now."""
        with self._lock:
            past = [n for n, _ in self.history]
            cur = self.cur
        done = sorted({k for prefix, k in self.FIRST_RUN_PATHS if any(n.startswith(prefix) for n in past)})
        inside = sorted({k for prefix, k in self.FIRST_RUN_PATHS if cur.startswith(prefix)})
        return {"completed": done, "in_progress": inside, "of": "DESIGN.md section 7, rows 1-11 (11 = the final barrier, after the line)"}

    # ---- failure record ----
    def record(self, error: str, kind: str) -> dict:
        now = time.monotonic()
        rec = dict(self.base)
        rec.update({"value": None, "error": error, "failure": kind, "stage": self.cur, "stage_elapsed_s": round(now - self.cur_t0, 2),
                    "elapsed_s": round(now - self.t0, 2), "rank": self.rank, "n_gpus": self.base.get("n_gpus", self.world),
                    "host": socket.gethostname(), "stages_done": [f"{n}:{t}s" for n, t in self.history[-12:]]})
        if self.world > 1:
            rec["first_run_paths"] = self.first_run_paths()
        return rec

    def _peer_records(self, wait: float) -> list[dict]:
        """Records other ranks left in the shared directory (the rank that failed FIRST holds the cause)."""
        end = time.monotonic() + wait
        out: dict[str, dict] = {}
        while True:
            try:
                for fn in os.listdir(self.share_dir):
                    if fn.startswith("fail_rank") and fn not in out and fn != f"fail_rank{self.rank}.json":
                        try:
                            if os.path.getmtime(os.path.join(self.share_dir, fn)) < self.wall0 - 1.0:
                                continue            # left by an earlier launch on the same port
                            with open(os.path.join(self.share_dir, fn)) as f:
                                out[fn] = json.load(f)
                        except Exception:
                            pass
            except FileNotFoundError:
                pass
            if out or time.monotonic() >= end:
                break
            time.sleep(0.1)
        return [{k: v.get(k) for k in ("rank", "stage", "failure", "error", "elapsed_s")} for v in out.values()]

    def fail(self, error: str, kind: str, code: int) -> None:
        """Leave the record, print it, end the process.  Safe to call from any thread; only the first call acts."""
        with self._lock:
            if self._done:
                return
            second = self._failed
            self._failed = True
        if second:              # another thread is already writing the record and will end the process: do not race it to the exit
            while True:
                time.sleep(1.0)
        rec = self.record(error, kind)
        try:
            os.makedirs(self.share_dir, mode=0o700, exist_ok=True)
            if os.path.islink(self.share_dir) or os.stat(self.share_dir).st_uid != os.getuid():
                raise PermissionError(f"{self.share_dir} is not ours")
            tmp = os.path.join(self.share_dir, f".fail_rank{self.rank}.tmp")
            with open(tmp, "w") as f:
                json.dump(rec, f)
            os.replace(tmp, os.path.join(self.share_dir, f"fail_rank{self.rank}.json"))
        except Exception:
            pass
        if self.rank == self.stdout_rank:
            # terminated / timed out because of someone else: give the culprit a moment to leave its record
            rec["peer_failures"] = self._peer_records(2.0 if kind != "exception" else 0.3)
            try:
                sys.stdout.write(json.dumps(rec) + "\n")
                sys.stdout.flush()
            except Exception:
                pass
        try:
            sys.stderr.write(f"[ssd_amd watchdog] rank {self.rank}: {kind} in stage '{rec['stage']}': {error}\n")
            sys.stderr.flush()
        except Exception:
            pass
        os._exit(code)

    # ---- threads ----
    def _watch(self) -> None:
        while not self._done:
            time.sleep(0.25)
            now = time.monotonic()
            if self._done:
                return
            if now > self.cur_deadline:
                self.fail(f"stage '{self.cur}' exceeded its {self.cur_deadline - self.cur_t0:.0f} s limit", "timeout", self.EXIT_TIMEOUT)
            if now - self.t0 > self.total_deadline:
                self.fail(f"run exceeded its total limit of {self.total_deadline:.0f} s", "timeout", self.EXIT_TIMEOUT)

    def _signals(self) -> None:
        while not self._done:
            try:
                data = os.read(self._rfd, 16)
            except OSError:
                return
            if not data:
                return
            if any(b in (signal.SIGTERM, signal.SIGHUP) for b in data) and not self._done:
                self.fail("terminated by the launcher (SIGTERM): another rank failed first or the launch was cancelled",
                          "terminated", self.EXIT_TERMINATED)

    def install(self) -> "RunGuard":
        global _ACTIVE
        _ACTIVE = self
        try:
            os.remove(os.path.join(self.share_dir, f"fail_rank{self.rank}.json"))
        except OSError:
            pass
        threading.Thread(target=self._watch, name="ssd-watchdog", daemon=True).start()
        if threading.current_thread() is threading.main_thread():
            try:
                self._rfd, self._wfd = os.pipe()
                os.set_blocking(self._wfd, False)
                signal.set_wakeup_fd(self._wfd, warn_on_full_buffer=False)
                # a Python-level handler must exist for the C handler (which feeds the wake-up fd) to be installed; it does
                # nothing itself: the reader thread acts, whether or not the main thread is stuck in a collective
                signal.signal(signal.SIGTERM, lambda *_: None)
                threading.Thread(target=self._signals, name="ssd-sigterm", daemon=True).start()
            except Exception:
                pass
        return self

    def run(self, fn):
        """fn() under the guard: an exception becomes a failure record + exit code, never a hang in interpreter shutdown
        (destructors of process groups / communicators can block for minutes once a peer is gone)."""
        try:
            out = fn()
        except SystemExit:
            raise
        except BaseException as e:          # noqa: BLE001 -- every failure must end in a record
            tb = traceback.format_exc(limit=6)
            sys.stderr.write(tb)
            self.fail(f"{type(e).__name__}: {e}", "exception", self.EXIT_EXCEPTION)
            raise
        self.done()
        return out
