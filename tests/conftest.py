import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "bench: a timing comparison (wide band), kept apart from the functional assertions")


@pytest.fixture(scope="session")
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def suite_arithmetic():
    """HELLO_TEST_ARITHMETIC=bf16x3 | bf16x3+32 runs the WHOLE suite in that arithmetic: engines the tests create without
    naming one get it where the mode exists (the canonical fused read convolver) and exact fp32 elsewhere.  This lives in the
    test harness, not in hello_amd.engine: nothing in a user's environment can change what an Engine computes."""
    import os
    mode = os.environ.get("HELLO_TEST_ARITHMETIC", "fp32")
    if mode == "fp32":
        yield "fp32"
        return
    from hello_amd import compiler, engine
    plain = engine.Engine.__init__

    def init(self, spec, state, device=0, fused=True, winograd=True, program=None, arithmetic=None, lanes_max_sites=None):
        if program is None and arithmetic is None:
            try:
                program = compiler.compile_model(spec, state, fused=fused, winograd=winograd, arithmetic=mode)
            except ValueError:
                program = None          # the mode does not exist for this model / these options: exact fp32
        plain(self, spec, state, device=device, fused=fused, winograd=winograd, program=program, arithmetic=arithmetic, lanes_max_sites=lanes_max_sites)
    engine.Engine.__init__ = init
    try:
        yield mode
    finally:
        engine.Engine.__init__ = plain
