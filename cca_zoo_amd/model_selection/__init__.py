"""Model selection for multiview models (surface of cca_zoo/model_selection/__init__.py)."""

from cca_zoo_amd.model_selection._search import GridSearchCV

__all__ = ["GridSearchCV"]
