from .base_model import BaseModel
