"""joshupscale_amd/provenance.py: the kernel -> source-file table bench.py's `roofline.traffic` guard rests on (CPU)."""

import os
import re

from joshupscale_amd import provenance as P


def test_every_listed_kernel_is_defined_in_the_file_the_table_names():
    """A PMC summary is dropped when the kernel's source changed since it was collected -- which only works when the
    digest covers the file that DEFINES the kernel (advisor, round 5: two kernels were mapped to a file that only
    mentions them in comments)."""
    for kernel, src in P.KERNEL_SOURCES.items():
        text = open(os.path.join(P.CSRC, src)).read()
        text = re.sub(r"//[^\n]*", "", text)          # a mention in a comment is not a definition
        assert re.search(r"__global__[^;{]*?\b" + re.escape(kernel) + r"\s*\(", text, re.S), (kernel, src)
        assert P.kernel_source_digest(kernel) is not None
        assert P.kernel_source_digest(kernel + "<stream>") == P.kernel_source_digest(kernel)
    assert P.kernel_source_digest("no_such_kernel") is None
