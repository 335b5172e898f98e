"""Device build of csrc/special.h against mpmath."""
import pytest

from util_special import check_all

pytestmark = pytest.mark.gpu


def test_special_functions_device_build():
    check_all(device=True)
