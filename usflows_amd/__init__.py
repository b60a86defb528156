"""usflows_amd -- MI355X-native implementation of the USFlows coupling-flow hot path
(Flow.log_prob / Flow.sample) behind the reference's Flow / Transform nn.Module API."""
__version__ = "0.1.0"
