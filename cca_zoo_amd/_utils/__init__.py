"""Validation + linear-algebra seams (device-backed)."""

from cca_zoo_amd._utils._linalg import gevp, svd_whiten
from cca_zoo_amd._utils._validation import perview_parameter, validate_views

__all__ = ["gevp", "svd_whiten", "perview_parameter", "validate_views"]
