"""CPU: the per-table reader/writer lock of the VecDB mirror (reference: RwLock per table, database/mod.rs:248-256)."""
import threading
import time

from lab_1806_vec_db_amd.vecdb import _RwLock


def test_readers_share_writers_exclude():
    lk = _RwLock()
    state = {"readers": 0, "max_readers": 0, "writer": False, "bad": 0}
    mu = threading.Lock()

    def reader():
        for _ in range(200):
            with lk.read():
                with mu:
                    state["readers"] += 1
                    state["max_readers"] = max(state["max_readers"], state["readers"])
                    state["bad"] += state["writer"]
                time.sleep(0.0002)
                with mu:
                    state["readers"] -= 1

    def writer():
        for _ in range(50):
            with lk.write():
                with mu:
                    state["bad"] += state["writer"] or state["readers"] > 0
                    state["writer"] = True
                time.sleep(0.0005)
                with mu:
                    state["writer"] = False
            time.sleep(0.001)

    ts = [threading.Thread(target=reader) for _ in range(4)] + [threading.Thread(target=writer) for _ in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert state["bad"] == 0
    assert state["max_readers"] > 1  # reads overlapped each other


def test_waiting_writer_holds_back_new_readers():
    lk = _RwLock()
    order = []
    r_in, w_waiting = threading.Event(), threading.Event()

    def first_reader():
        with lk.read():
            r_in.set()
            w_waiting.wait(2)
            time.sleep(0.05)
            order.append("r1")

    def the_writer():
        r_in.wait(2)
        w_waiting.set()
        with lk.write():
            order.append("w")

    def late_reader():
        w_waiting.wait(2)
        time.sleep(0.02)  # arrives while the writer waits for r1
        with lk.read():
            order.append("r2")

    ts = [threading.Thread(target=f) for f in (first_reader, the_writer, late_reader)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    assert order == ["r1", "w", "r2"], order
