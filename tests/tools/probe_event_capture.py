"""Probe (GPU box): when does hipEventQuery return hipErrorCapturedEvent?

The abort of round 2 (ProcessGroupNCCL's watchdog thread: WorkNCCL::isCompleted -> event query -> hipErrorCapturedEvent)
needs an event "last recorded in a capturing stream".  torch never enqueues a Work created under capture, so which
event was that?  This probe records an event EAGERLY on a stream, lets it complete, then starts a capture that the
stream joins, and queries the event from a second thread (as the watchdog does):

  case A  event recorded eagerly on stream S, S idle, no capture anywhere        -> query OK
  case B  same event, while ANOTHER stream captures (S not involved)             -> query OK ?
  case C  same event, while S itself is part of a capture (joined by an event)   -> query fails ?

If C fails and B does not, the rule on this HIP is "the stream the event was last recorded on is capturing NOW",
not "the record was captured": an un-retired eager Work of a process group whose internal stream later joins a capture
is enough to kill the watchdog.  Output: one JSON line."""
import json
import threading

import torch


def query_in_thread(ev):
    out = {}

    def go():
        try:
            out["done"] = bool(ev.query())
        except Exception as e:                                  # noqa: BLE001
            out["error"] = str(e).split("\n")[0][:160]
    th = threading.Thread(target=go)
    th.start(); th.join()
    return out


def main():
    torch.cuda.set_device(0)
    s = torch.cuda.Stream()
    other = torch.cuda.Stream()
    x = torch.ones(1 << 20, device="cuda")
    ev = torch.cuda.Event()
    with torch.cuda.stream(s):
        x.mul_(2.0)
        ev.record(s)
    torch.cuda.synchronize()
    res = {"A_idle": query_in_thread(ev)}
    print(json.dumps(res), flush=True)
    try:
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=other, capture_error_mode="thread_local"):
            x.add_(1.0)
            res["B_other_stream_capturing"] = query_in_thread(ev)
    except Exception as e:                                      # noqa: BLE001
        res["B_capture_error"] = str(e).split("\n")[0][:160]
    print(json.dumps(res), flush=True)
    try:
        g2 = torch.cuda.CUDAGraph()
        fork, join = torch.cuda.Event(), torch.cuda.Event()
        with torch.cuda.graph(g2, stream=other, capture_error_mode="thread_local"):
            x.add_(1.0)
            fork.record(other)
            s.wait_event(fork)                                  # S joins the capture (what ncclStream does for a captured collective)
            with torch.cuda.stream(s):
                x.mul_(1.0)
                join.record(s)
            other.wait_event(join)
            x.add_(1.0)
            res["C_its_stream_joined_the_capture"] = query_in_thread(ev)
    except Exception as e:                                      # noqa: BLE001
        res["C_capture_error"] = str(e).split("\n")[0][:160]
    print(json.dumps(res), flush=True)
    torch.cuda.synchronize()
    res["D_after_capture"] = query_in_thread(ev)
    print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
