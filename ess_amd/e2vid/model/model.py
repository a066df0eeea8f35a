"""
E2VID model classes on the HIP kernels (reference: e2vid/model/model.py).

The config dict is the one stored in E2VID checkpoints: `num_bins` is mandatory, the rest default as in
BaseE2VID (model.py:9-44): skip_type 'sum', num_encoders 4, base_num_channels 32, num_residual_blocks 2,
norm None, use_upsample_conv True, recurrent_block_type 'convlstm'.
"""
from ..base import BaseModel
from .unet import UNet, UNetDecoder, UNetRecurrent, UNetTask


class BaseE2VID(BaseModel):
    _DEFAULTS = (('skip_type', str, 'sum'), ('num_encoders', int, 4), ('base_num_channels', int, 32),
                 ('num_residual_blocks', int, 2), ('norm', str, None), ('use_upsample_conv', bool, True))

    def __init__(self, config):
        super().__init__(config)
        assert 'num_bins' in config
        self.num_bins = int(config['num_bins'])
        for name, cast, default in self._DEFAULTS:
            setattr(self, name, cast(config[name]) if name in config else default)
        # the reference stringifies whatever is stored under 'norm' (model.py:36): None -> 'None' = no norm
        self.recurrent_block_type = str(config['recurrent_block_type']) if 'recurrent_block_type' in config else 'convlstm'

    def _unet_kwargs(self):
        return dict(num_input_channels=self.num_bins, num_output_channels=1, skip_type=self.skip_type,
                    activation='sigmoid', num_encoders=self.num_encoders, base_num_channels=self.base_num_channels,
                    num_residual_blocks=self.num_residual_blocks, norm=self.norm,
                    use_upsample_conv=self.use_upsample_conv)


class E2VID(BaseE2VID):
    def __init__(self, config):
        super().__init__(config)
        self.unet = UNet(**self._unet_kwargs())

    def forward(self, event_tensor, prev_states=None):
        """event_tensor N x num_bins x H x W -> (image N x 1 x H x W in [0,1], None)."""
        return self.unet.forward(event_tensor), None


class E2VIDRecurrent(BaseE2VID):
    """The ESS event encoder: recurrent UNet with a ConvLSTM / ConvGRU after every encoder (model.py:69-100)."""

    def __init__(self, config):
        super().__init__(config)
        self.unetrecurrent = UNetRecurrent(recurrent_block_type=self.recurrent_block_type, **self._unet_kwargs())

    def forward(self, event_tensor, prev_states, encoder_only=False, lean=False, prefix=None, lean_state=False):
        """-> (img N x 1 x H x W, states per encoder, latent {1,2,4,8}); encoder_only / lean / prefix / lean_state: see UNetRecurrent.forward."""
        return self.unetrecurrent.forward(event_tensor, prev_states, encoder_only=encoder_only, lean=lean, prefix=prefix, lean_state=lean_state)

    def forward_prefix(self, event_tensors):
        """head + first encoder conv for many time slices at once (UNetRecurrent.forward_prefix)."""
        return self.unetrecurrent.forward_prefix(event_tensors)


class E2VIDDecoder(BaseE2VID):
    def __init__(self, config):
        super().__init__(config)
        self.unetrecurrent = UNetDecoder(recurrent_block_type=self.recurrent_block_type, **self._unet_kwargs())

    def forward(self, x, blocks, head):
        return self.unetrecurrent.forward(x, blocks, head)


class E2VIDTask(BaseE2VID):
    """E2VID's decoder half with a 13-class semantic head on the latents (reference model.py:135-166)."""

    def __init__(self, config):
        super().__init__(config)
        kw = self._unet_kwargs()
        kw['num_output_channels'] = 13
        self.unetrecurrent = UNetTask(recurrent_block_type=self.recurrent_block_type, **kw)

    def forward(self, input_dict):
        return self.unetrecurrent.forward(input_dict)


ARCHS = {'E2VID': E2VID, 'E2VIDRecurrent': E2VIDRecurrent, 'E2VIDDecoder': E2VIDDecoder, 'E2VIDTask': E2VIDTask}
