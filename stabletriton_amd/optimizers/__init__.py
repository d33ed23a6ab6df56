from .remove_dropout import remove_dropout
from .replace_attention import fuse_attention
from .replace_geglu import fuse_geglu
from .replace_groupnorm import replace_group_norm, replace_group_norm_activation
from .replace_layernorm import replace_layer_norm
from .replace_linear import replace_linear, replace_linear_activ
from .replace_conv import replace_conv
from .replace_timesteps import fuse_timesteps
from .fuse_epilogues import fuse_geglu_into_linear, fuse_residual_adds, fuse_temb_add
from .fuse_projections import (fuse_layernorm_into_linear, fuse_query_projection_into_attention, fuse_shared_input_linears, split_context,
                               split_region)
from .plan_fp8 import plan_fp8
from .fuse_groupnorm_stats import fuse_groupnorm_stats
from .fuse_skip_cat import fuse_skip_cat
from .cleanup import dedupe_pure_calls, fuse_token_residual
from .layout import keep_channels_last
from .graphs import make_dynamic_graphed_callable
from . import wrappers
