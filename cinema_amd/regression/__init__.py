"""Regression fine-tuning of ConvViT on the HIP path (interface of the reference ``cinema/regression``)."""
