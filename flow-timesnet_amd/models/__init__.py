"""Drop-in mirrors of the reference's ``timesnet_forecast.models`` modules."""
from . import timesnet  # noqa: F401
from .timesnet import (  # noqa: F401
    FFTPeriodSelector, InceptionBlock, InceptionBranch, LowRankTemporalContext, PeriodGrouper,
    PeriodGroupResult, TimesBlock,
)
from .shell import DataEmbedding, PositionalEmbedding, RMSNorm, TimesNet  # noqa: E402,F401
