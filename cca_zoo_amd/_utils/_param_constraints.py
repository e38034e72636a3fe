"""sklearn ``_parameter_constraints`` fragments shared by the estimators.

Same accepted ranges as the reference (cca_zoo/_utils/_param_constraints.py:18-25)
so that ``InvalidParameterError`` fires for the same inputs.
"""

from __future__ import annotations

from numbers import Integral, Real
from typing import Any

from sklearn.utils._param_validation import Interval

RIDGE_PARAMETER: list[Any] = [Interval(Real, 0, 1, closed="both"), "array-like"]
POSITIVE_EPS: list[Any] = [Interval(Real, 0, None, closed="neither")]
POSITIVE_INT: list[Any] = [Interval(Integral, 1, None, closed="left")]
