"""Two implementations behind one test body: the float64 CPU oracle (runs everywhere) and the HIP
product path (``-m gpu``).  Tests read like the reference's own tests; only tolerances differ
(float64: the reference's; fp32 HIP: stated per test)."""
import pytest
import torch


class _Oracle:
    name, device, dtype, is_hip = "oracle", "cpu", torch.float64, False

    def __init__(self):
        import oracle.ggn, oracle.lla, oracle.sample, oracle.stochtrace, oracle.matfree
        self.ggn, self.lla, self.sample, self.stochtrace, self.matfree = (
            oracle.ggn, oracle.lla, oracle.sample, oracle.stochtrace, oracle.matfree)

    def state(self, st):
        return st

    def tensor(self, t):
        return t.to(torch.float64)

    def rows(self, fun, M):          # apply an oracle to every row (the reference vmaps)
        return torch.stack([fun(m) for m in M])

    def tol(self, f64, f32):
        return f64


class _Hip:
    name, device, dtype, is_hip = "hip", "cuda", torch.float32, True

    def __init__(self):
        import src.ggn, src.lla, src.sample, src.stochtrace
        self.ggn, self.lla, self.sample, self.stochtrace = src.ggn, src.lla, src.sample, src.stochtrace

    def state(self, st):
        return st.to(device="cuda", dtype=torch.float32)

    def tensor(self, t):
        return t.to(device="cuda", dtype=torch.float32)

    def rows(self, fun, M):
        return fun.rows(M) if hasattr(fun, "rows") else torch.stack([fun(m) for m in M])

    def tol(self, f64, f32):
        return f32


@pytest.fixture(params=["oracle", pytest.param("hip", marks=pytest.mark.gpu)])
def impl(request):
    return _Oracle() if request.param == "oracle" else _Hip()


def cpu64(t):
    return t.detach().to("cpu", torch.float64)
