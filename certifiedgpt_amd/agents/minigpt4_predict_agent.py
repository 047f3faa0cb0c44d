"""Predict agent (launch.py:94-96 mode `smoothing_predict`; the reference's agents/minigpt4_predict_agent.py is empty):
the certify loop with Smooth.predict (smoothing.py:58-79) instead of certify."""
from .minigpt4_certify_agent import MiniGPT4CertifyAgent
from .registry import registry


@registry.register_agent("image_text_predict")
class MiniGPT4PredictAgent(MiniGPT4CertifyAgent):
    mode = "predict"
