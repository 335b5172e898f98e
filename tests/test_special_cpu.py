"""Host build of csrc/special.h (same source as the device functions) against mpmath."""
from util_special import check_all


def test_special_functions_host_build():
    check_all(device=False)
