"""Masked autoencoder (interface of the reference ``cinema/mae``)."""
